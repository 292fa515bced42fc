"""MI355X-native engine behind wayverb's `waveguide::run`.

  engine      ctypes binding of the C ABI (include/wayverb_amd.h): Engine, run / run_fast, SceneMesh
  mesh        the reference's data contract (condensed_node, boundary_data, coefficients) + test meshes
  slab        z-slab decomposition across ranks
  scene       triangle scenes (OBJ reader, generators), adjusted boundary
  wayfile     wayverb `.way` project bundles (config.json + model.model)
  filters     wall filter design (host C++ in the library)
  postprocess receiver traces -> audio
  simulation  compute_voxels_and_mesh / canonical / impulse_response
  build       hipcc build of libwayverb_amd.so (gfx950)
"""
