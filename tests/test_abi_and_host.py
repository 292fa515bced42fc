"""CPU: the C-ABI library loads and exports every symbol include/wayverb_amd.h declares; host
helpers; the product fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from wayverb_amd import mesh as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "wayverb_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_library):
    lib = ctypes.CDLL(built_library)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libwayverb_amd.so does not export %s" % n
    from wayverb_amd import engine as E
    assert sorted(E.EXPORTS) == names


def test_struct_layouts_match_the_reference_contract():
    from wayverb_amd import engine as E
    assert M.condensed_node_dtype.itemsize == 8 and M.boundary_data_dtype.itemsize == 56
    assert M.coefficients_dtype.itemsize == 112
    # the ctypes mirrors against the header itself, as a C compiler lays it out
    import subprocess
    import tempfile
    fields = ["struct_size", "precision", "device", "ghost_lo", "ghost_hi", "flag_interval", "stream_variant", "all_tiles",
              "nodes_on_device", "comm_timeout_s", "transport", "tuning"]
    tfields = list(E.TUNING_FIELDS)
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"wayverb_amd.h\"\nint main(void){printf(\"%zu %zu\", sizeof(wv_options), sizeof(wv_tuning));" + \
        "".join('printf(" %%zu", offsetof(wv_options, %s));' % f for f in fields) + \
        "".join('printf(" %%zu", offsetof(wv_tuning, %s));' % f for f in tfields) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as tmp:
        src, exe = os.path.join(tmp, "layout.c"), os.path.join(tmp, "layout")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    want = [ctypes.sizeof(E.WvOptions), ctypes.sizeof(E.WvTuning)] + [getattr(E.WvOptions, f).offset for f in fields] + \
        [getattr(E.WvTuning, f).offset for f in tfields]
    assert got == want
    assert ctypes.sizeof(E.WvOptions) == 64 + ctypes.sizeof(E.WvTuning)  # (64: the struct before wv_tuning was appended)


def test_box_mesh_counts_match_survey_table():
    m = M.box_mesh(32, 32, 32)
    t = m.nodes["boundary_type"]
    assert int((t == M.ID_INSIDE).sum()) == 21952 and int((t == 0).sum()) == 5768
    assert [b.shape[0] for b in m.bidx] == [4704, 336, 8]


def test_native_box_helper_matches_numpy_and_slabs(built_library):
    from wayverb_amd import engine as E
    nodes, counts = E.make_box_nodes(13, 9, 11)
    ref = M.box_mesh(13, 9, 11)
    assert nodes.tobytes() == ref.nodes.tobytes()
    assert counts == tuple(b.shape[0] for b in ref.bidx)
    # a slab [3, 8) with ghosts [2, 9): types global, numbering restarts over owned planes
    slab, c = E.make_box_nodes(13, 9, 11, z_begin=2, z_count=7, number_from=3, number_to=8)
    plane = 13 * 9
    assert np.array_equal(slab["boundary_type"], ref.nodes["boundary_type"][2 * plane:9 * plane])
    owned = slab[plane:6 * plane]
    for d in (1, 2, 3):
        sel = np.array([bin(int(t)).count("1") == d and not (t & 1) for t in owned["boundary_type"]])
        assert np.array_equal(owned["boundary_index"][sel], np.arange(sel.sum()))
        assert c[d - 1] == sel.sum()


def test_no_cpu_fallback(built_library):
    """Without a HIP device wv_create must fail loudly (never compute on the host)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from wayverb_amd import engine as E
    with pytest.raises(E.WaveguideError, match="(?i)no HIP device|hip"):
        E.Engine(M.box_mesh(8, 8, 8))


def test_product_does_not_touch_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "wayverb_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "oracle" not in text.replace("no CPU fallback", ""), f


def test_box_boundary_row_arithmetic(built_library):
    """helpers.box_boundary_rows_below (used by the full-size 1024^3 GPU test to find a plane's rows in
    the boundary arrays without building the 8.6 GB node array twice) against wv_make_box_nodes."""
    from helpers import box_boundary_rows_below
    from wayverb_amd import engine as E
    nx, ny, nz = 11, 9, 13
    for z in range(nz + 1):
        _, counts = E.make_box_nodes(nx, ny, nz, z_begin=0, z_count=max(z, 1), number_from=0, number_to=z)
        assert [box_boundary_rows_below(nx, ny, nz, z, d) for d in (1, 2, 3)] == list(counts)


def test_the_library_reads_no_environment_variables():
    """Product behaviour is steered through wv_options / wv_tuning only; the one getenv in the sources sits in the
    WV_DEBUG_ENV block of engine.hip, which the product build does not define."""
    csrc = os.path.join(ROOT, "wayverb_amd", "csrc")
    hits = []
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".cpp")):
            text = open(os.path.join(csrc, name)).read()
            hits += [(name, m.start()) for m in re.finditer(r"\bgetenv\b", text)]
    assert [h[0] for h in hits] == ["engine.hip"]
    text = open(os.path.join(csrc, "engine.hip")).read()
    block = text[text.index("#ifdef WV_DEBUG_ENV"):]
    block = block[:block.index("#endif")]
    assert hits[0][1] > text.index("#ifdef WV_DEBUG_ENV") and "getenv" in block
    from wayverb_amd import build as B
    assert not any("WV_DEBUG_ENV" in f for f in B.FLAGS)


def test_tuning_reaches_the_options_struct_and_unknown_fields_are_refused(built_library):
    """Engine tuning travels in wv_options::tuning: the Python side fills the struct from a dict (engine.default_tuning
    merged with the call's own), refuses names the header does not have, and tools may read WV_* names from their own
    environment -- the library never does."""
    from wayverb_amd import engine as E
    lib = E.load_library()
    opt = E.WvOptions()
    lib.wv_default_options(ctypes.byref(opt))
    assert opt.struct_size == ctypes.sizeof(E.WvOptions) and opt.precision == E.PRECISION_F64 and opt.stream_variant == 2
    t = opt.tuning
    assert (t.pair, t.pair_inner_fix, t.pair_wide, t.pair_unit_planes, t.tile_lists, t.fuse_pre_post, t.graph, t.boundary_lds,
            t.boundary_order, t.boundary_xwall, t.slab_early, t.pair_split_rows, t.fuse_planes, t.whole_step) == (-1, 1, 1, 32, 1, 1, 0, 1, 1, 1, -1, 0, 1, -1)
    old = dict(E.default_tuning)
    try:
        E.default_tuning.clear()
        E.default_tuning.update(pair=0, stream_variant=3)
        E.apply_tuning(opt, dict(boundary_xwall=0, pair=1))
        assert (opt.tuning.pair, opt.tuning.boundary_xwall, opt.stream_variant) == (1, 0, 3)
        with pytest.raises(ValueError):
            E.apply_tuning(opt, dict(no_such_knob=1))
    finally:
        E.default_tuning.clear()
        E.default_tuning.update(old)
    assert E.tuning_from_env({"WV_PAIR": "0", "WV_STREAM_RY": "2", "WV_STREAM_VARIANT": "3", "HOME": "/"}) == \
        dict(pair=0, stream_ry=2, stream_variant=3)


def test_no_kernel_of_the_engine_spills(built_library):
    """The march runs on 255 of 256 VGPRs and used to carry 44 B of scratch per lane, worth 7 % of its time:
    the compiler's own account of engine.hip's kernels, written beside the library by wayverb_amd.build, must show no
    scratch in any of them, two waves per SIMD for the march and at least three for the boundary kernels (126-144 VGPRs)."""
    from wayverb_amd import build as B
    text = open(B.RESOURCES).read()
    blocks = re.split(r"remark: Function Name: ", text)[1:]
    assert len(blocks) > 50
    seen = set()
    for b in blocks:
        name = b.split()[0]
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        waves = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        assert scratch == 0, (name, scratch)
        if "pair_march_kernel" in name:
            assert waves >= 2, (name, waves)
            seen.add("march")
        if "boundary_kernel" in name:
            assert waves >= 3, (name, waves)
            seen.add("boundary")
    assert seen == {"march", "boundary"}


def test_bench_without_a_gpu_prints_an_error_line(built_library):
    """bench.py's contract is ONE JSON line whatever happens: without a GPU the line carries "error" and the exit status is 1."""
    import json
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 1 and len(lines) == 1, (p.stdout, p.stderr[-1500:])
    line = json.loads(lines[0])
    assert line["value"] is None and "needs a GPU" in line["error"] and line["n_gpus"] == 1
