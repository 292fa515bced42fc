#!/usr/bin/env python3
"""Worker process of oracle.oracle.ReferenceOnDevice (TEST / BENCH-BASELINE INFRASTRUCTURE ONLY).

ROCm's OpenCL runtime and the HIP runtime bundled with torch do not share a process well (two copies
of the HSA runtime), so the reference's OpenCL program runs in a process of its own that loads neither
torch nor the engine:

    python oracle/ref_cl_worker.py run   <in.npz> <out.npz>     one waveguide::run from arrays on disk
    python oracle/ref_cl_worker.py bench <n> <steps> <f32|f64>  n^3 box built here, timed; prints one JSON line
    python oracle/ref_cl_worker.py name                          prints the OpenCL GPU device name (or nothing)
    python oracle/ref_cl_worker.py setup <in.npz> <out.npz>     set_node_inside + set_node_boundary_type (mesh.cpp:75-111)
    python oracle/ref_cl_worker.py bcf   <in.npz> <out.npz>     the three boundary_coefficient_finder kernels
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def load():
    lib = C.CDLL(os.path.join(HERE, "_ref", "libwvref_cl.so"))
    lib.wvrefcl_last_error.restype = C.c_char_p
    lib.wvrefcl_available.argtypes = [C.c_char_p, C.c_int]
    lib.wvrefcl_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32,
                                C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64,
                                C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int),
                                C.POINTER(C.c_double)]
    lib.wvrefcl_mesh_setup.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_uint64,
                                       C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                       C.c_void_p]
    lib.wvrefcl_boundary_coefficient_finder.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                                        C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64,
                                                        C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    return lib


def run(lib, dims, nodes, coeffs, bd, previous, current, source_kind, source_node, signal, n_steps, recv, contract_off):
    dtype = previous.dtype
    nx, ny, nz = (int(v) for v in dims)
    sig = np.ascontiguousarray(signal, dtype=np.float64)
    rc = np.ascontiguousarray(recv, dtype=np.uint64)
    trace = np.zeros((n_steps, len(rc)), dtype=dtype)
    done, flag, secs = C.c_uint64(), C.c_int(), C.c_double()
    err = lib.wvrefcl_run(dtype.itemsize, int(contract_off), nx, ny, nz, _ptr(nodes), _ptr(coeffs), coeffs.shape[0],
                          _ptr(bd[0]), bd[0].shape[0], _ptr(bd[1]), bd[1].shape[0], _ptr(bd[2]), bd[2].shape[0],
                          _ptr(previous), _ptr(current), int(source_kind), int(source_node), _ptr(sig), int(n_steps),
                          _ptr(rc), len(rc), _ptr(trace), C.byref(done), C.byref(flag), C.byref(secs))
    if err:
        raise SystemExit("reference on the OpenCL device: " + lib.wvrefcl_last_error().decode())
    return int(done.value), int(flag.value), trace[:int(done.value)], float(secs.value)


def main():
    lib = load()
    mode = sys.argv[1]
    if mode == "name":
        buf = C.create_string_buffer(256)
        if lib.wvrefcl_available(buf, 256):
            print(buf.value.decode())
        return
    if mode == "run":
        d = np.load(sys.argv[2])
        prev, cur = d["previous"].copy(), d["current"].copy()
        bd = [d["bd1"].copy(), d["bd2"].copy(), d["bd3"].copy()]
        done, flag, trace, secs = run(lib, d["dims"], d["nodes"], d["coefficients"], bd, prev, cur, int(d["source_kind"]),
                                      int(d["source_node"]), d["signal"], int(d["n_steps"]), d["recv"], bool(d["contract_off"]))
        np.savez(sys.argv[3], steps=done, flag=flag, trace=trace, seconds=secs, previous=prev, current=cur,
                 bd1=bd[0], bd2=bd[1], bd3=bd[2])
        return
    if mode == "setup":
        d = np.load(sys.argv[2])
        nx, ny, nz = (int(v) for v in d["dims"])
        nodes = np.zeros(nx * ny * nz * 2, dtype=np.uint32)          # condensed_node[n], zero-filled
        mc = np.ascontiguousarray(d["min_corner"], dtype=np.float32)
        vox = np.ascontiguousarray(d["voxel_index"], dtype=np.uint32)
        a0, a1 = (np.ascontiguousarray(d[k], dtype=np.float32) for k in ("aabb_c0", "aabb_c1"))
        tri = np.ascontiguousarray(d["triangles"], dtype=np.uint32)
        ver = np.ascontiguousarray(d["vertices"], dtype=np.float32)
        err = lib.wvrefcl_mesh_setup(int(d["contract_off"]), nx, ny, nz, float(d["spacing"]), _ptr(mc), _ptr(vox), vox.shape[0],
                                     _ptr(a0), _ptr(a1), int(d["side"]), _ptr(tri), tri.shape[0], _ptr(ver), ver.shape[0], _ptr(nodes))
        if err:
            raise SystemExit("reference set-up on the OpenCL device: " + lib.wvrefcl_last_error().decode())
        np.savez(sys.argv[3], nodes=nodes)
        return
    if mode == "bcf":
        d = np.load(sys.argv[2])
        nx, ny, nz = (int(v) for v in d["dims"])
        mc = np.ascontiguousarray(d["min_corner"], dtype=np.float32)
        nodes = np.ascontiguousarray(d["nodes"])
        tri = np.ascontiguousarray(d["triangles"], dtype=np.uint32)
        ver = np.ascontiguousarray(d["vertices"], dtype=np.float32)
        n1, n2, n3 = (int(v) for v in d["counts"])
        o1, o2, o3 = np.zeros(max(n1, 1), np.uint32), np.zeros((max(n2, 1), 2), np.uint32), np.zeros((max(n3, 1), 3), np.uint32)
        err = lib.wvrefcl_boundary_coefficient_finder(int(d["contract_off"]), nx, ny, nz, float(d["spacing"]), _ptr(mc), _ptr(nodes),
                                                      _ptr(tri), tri.shape[0], _ptr(ver), ver.shape[0], _ptr(o1), n1, _ptr(o2), n2,
                                                      _ptr(o3), n3)
        if err:
            raise SystemExit("reference boundary coefficient finder on the OpenCL device: " + lib.wvrefcl_last_error().decode())
        np.savez(sys.argv[3], b1=o1[:n1], b2=o2[:n2], b3=o3[:n3])
        return
    if mode == "bench":
        from wayverb_amd import mesh as M          # pure numpy: no engine, no torch
        n, steps, tag = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
        dtype = np.float32 if tag == "f32" else np.float64
        mesh = M.box_mesh(n, n, n, coefficients=M.bench_materials(), surface_of_face=[0, 1, 2, 3, 2, 3])
        bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
        prev = np.zeros(mesh.num_nodes, dtype=dtype)
        cur = np.zeros(mesh.num_nodes, dtype=dtype)
        sig = np.zeros(steps)
        sig[0] = 1.0
        src = mesh.compute_index(n // 2, n // 2, n // 2)
        buf = C.create_string_buffer(256)
        lib.wvrefcl_available(buf, 256)
        run(lib, mesh.dims, mesh.nodes, mesh.coefficients, bd, prev, cur, 1, src, sig[:3], 3, [src + 3], False)   # warm-up (JIT)
        done, flag, trace, secs = run(lib, mesh.dims, mesh.nodes, mesh.coefficients, bd, prev, cur, 1, src, sig, steps,
                                      [src + 3], False)
        print(json.dumps({"value": round(mesh.num_nodes * done / secs / 1e9, 4), "unit": "Gnode-updates/s",
                          "device": buf.value.decode(), "steps": done, "flag": flag, "seconds": round(secs, 4),
                          "mesh": "%d^3 %s box" % (n, tag),
                          "what": "the reference's OpenCL program (condensed_waveguide, as written: 8-byte condensed_node "
                                  "per node, one work-item per node, per-step host round trips for the flag / source / "
                                  "receiver) compiled by this device's OpenCL runtime, host loop of waveguide.h:43-123"}))
        return
    raise SystemExit("unknown mode " + mode)


if __name__ == "__main__":
    main()
