"""ctypes bindings for the TEST-ONLY checkers in oracle/.

  Oracle()          -> liboracle.so   (the repo's C restatement, oracle/waveguide_oracle.c)
  Reference("f32")  -> _ref/libwvref_f32.so  (reference kernel text compiled for the host)
  Reference("f64")  -> _ref/libwvref_f64.so  (same, pressure type promoted to double)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def build_oracle():
    so = os.path.join(HERE, "liboracle.so")
    srcs = [os.path.join(HERE, f) for f in ("waveguide_oracle.c", "waveguide_oracle_body.h", "mesh_setup_oracle.c",
                                          "node_inside_oracle.c", "boundary_surfaces_oracle.c")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        for sfx in ("f32", "f64"):
            f = getattr(self.lib, "wvo_step_" + sfx)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            h = getattr(self.lib, "wvo_step_range_" + sfx)
            h.restype = C.c_int
            h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
            g = getattr(self.lib, "wvo_run_" + sfx)
            g.restype = C.c_int64
            g.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                          C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                          C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        self.lib.wvo_filter_test_2.argtypes = [C.c_void_p] * 4 + [C.c_int]
        self.lib.wvo_filter_test.argtypes = [C.c_void_p] * 4 + [C.c_int]
        self.lib.wvo_directional_receiver.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                                      C.c_double, C.c_void_p]

    def classify(self, inside_mask):
        """Node classification from an inside mask [nz, ny, nx] (bool): boundary types
        (`set_node_boundary_type`) + boundary indices.  Returns (nodes, (n1, n2, n3))."""
        from wayverb_amd import mesh as M
        nz, ny, nx = inside_mask.shape
        nodes = np.zeros(nx * ny * nz, dtype=M.condensed_node_dtype)
        nodes["boundary_type"] = np.where(inside_mask.reshape(-1), 1, 0)
        self.lib.wvo_set_node_boundary_type.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        self.lib.wvo_set_node_boundary_type(_ptr(nodes), nx, ny, nz)
        counts = (C.c_uint64 * 3)()
        self.lib.wvo_set_boundary_indices.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        self.lib.wvo_set_boundary_indices(_ptr(nodes), nodes.shape[0], counts)
        return nodes, tuple(int(c) for c in counts)

    def nodes_inside(self, dims, min_corner, spacing, voxel_index, aabb, side, triangles, vertices):
        """`set_node_inside` restated: uint8 mask [nz, ny, nx]."""
        nx, ny, nz = dims
        out = np.zeros(nx * ny * nz, dtype=np.uint8)
        mc = np.ascontiguousarray(min_corner, dtype=np.float32)
        a0 = np.ascontiguousarray(aabb[0], dtype=np.float32)
        a1 = np.ascontiguousarray(aabb[1], dtype=np.float32)
        self.lib.wvo_nodes_inside.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.wvo_nodes_inside.restype = None
        self.lib.wvo_nodes_inside(nx, ny, nz, _ptr(mc), float(spacing), _ptr(voxel_index), _ptr(a0), _ptr(a1), side,
                                  _ptr(triangles), _ptr(vertices), _ptr(out))
        return out.reshape(nz, ny, nx)

    def point_triangle_dist2(self, v0, v1, v2, p):
        f = self.lib.wvo_point_triangle_dist2
        f.restype = C.c_float
        f.argtypes = [C.c_void_p] * 4
        a = [np.ascontiguousarray(v, dtype=np.float32) for v in (v0, v1, v2, p)]
        return np.float32(f(*[_ptr(v) for v in a]))

    def boundary_coefficient_finder(self, nodes, dims, min_corner, spacing, triangles, vertices, counts,
                                    entry0_last_writer=False):
        """The three finder kernels restated, on first-numbering nodes: (out1 [n1], out2 [n2,2], out3 [n3,3])."""
        nx, ny, nz = dims
        mc = np.ascontiguousarray(min_corner, dtype=np.float32)
        o = [np.zeros((counts[d], d + 1), dtype=np.uint32) for d in range(3)]
        f = self.lib.wvo_boundary_coefficient_finder
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                      C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        f(_ptr(nodes), nx, ny, nz, float(spacing), _ptr(mc), _ptr(triangles), triangles.shape[0], _ptr(vertices),
          _ptr(o[0]), counts[0], _ptr(o[1]), counts[1], _ptr(o[2]), counts[2], int(entry0_last_writer))
        return o

    def boundary_index_data(self, nodes, dims, min_corner, spacing, triangles, vertices, entry0_last_writer=False):
        """compute_boundary_index_data restated.  nodes: types set (indices rewritten in place).
        Returns [b1 [n1,1], b2 [n2,2], b3 [n3,3]] surface indices."""
        nx, ny, nz = dims
        mc = np.ascontiguousarray(min_corner, dtype=np.float32)
        n = nodes.shape[0]
        o = [np.zeros((n, 1), dtype=np.uint32), np.zeros((n, 2), dtype=np.uint32), np.zeros((n, 3), dtype=np.uint32)]
        counts = (C.c_uint64 * 3)()
        f = self.lib.wvo_boundary_index_data
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        f(_ptr(nodes), nx, ny, nz, float(spacing), _ptr(mc), _ptr(triangles), triangles.shape[0], _ptr(vertices),
          _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), counts, int(entry0_last_writer))
        return [o[d][:int(counts[d])].copy() for d in range(3)]

    @staticmethod
    def real(dtype):
        dtype = np.dtype(dtype)
        return "f32" if dtype == np.float32 else "f64"

    def step(self, previous, current, mesh, bd, threads=1):
        """previous <- next in place. bd = [bd1, bd2, bd3] boundary_data arrays. Returns flag."""
        nx, ny, nz = mesh.dims
        f = getattr(self.lib, "wvo_step_" + self.real(previous.dtype))
        assert previous.dtype == current.dtype and previous.flags.c_contiguous
        return f(_ptr(previous), _ptr(current), _ptr(mesh.nodes), nx, ny, nz,
                 _ptr(bd[0]), _ptr(bd[1]), _ptr(bd[2]), _ptr(mesh.coefficients), threads)

    def step_range(self, previous, current, mesh, bd, z_begin, z_end, threads=1):
        """Like step(), restricted to planes [z_begin, z_end) (z-slab tests)."""
        nx, ny, nz = mesh.dims
        f = getattr(self.lib, "wvo_step_range_" + self.real(previous.dtype))
        return f(_ptr(previous), _ptr(current), _ptr(mesh.nodes), nx, ny, nz,
                 _ptr(bd[0]), _ptr(bd[1]), _ptr(bd[2]), _ptr(mesh.coefficients), z_begin, z_end, threads)

    def run(self, buf0, buf1, mesh, bd, source_kind, source_node, signal, n_steps, recv, threads=1):
        """The run loop. Returns (steps_completed, flag, out[steps, n_recv])."""
        nx, ny, nz = mesh.dims
        sfx = self.real(buf0.dtype)
        recv = np.ascontiguousarray(recv, dtype=np.int64)
        signal = np.ascontiguousarray(signal, dtype=np.float64) if signal is not None else None
        out = np.zeros((n_steps, len(recv)), dtype=buf0.dtype)
        flag = C.c_int(0)
        steps = getattr(self.lib, "wvo_run_" + sfx)(
            _ptr(buf0), _ptr(buf1), _ptr(mesh.nodes), nx, ny, nz,
            _ptr(bd[0]), _ptr(bd[1]), _ptr(bd[2]), _ptr(mesh.coefficients),
            source_kind, source_node, _ptr(signal), n_steps, _ptr(recv), len(recv),
            _ptr(out), threads, C.byref(flag))
        return int(steps), int(flag.value), out

    def filter_test_2(self, inp, memory, coeffs):
        inp = np.ascontiguousarray(inp, dtype=np.float32)
        out = np.zeros_like(inp)
        self.lib.wvo_filter_test_2(_ptr(inp), _ptr(out), _ptr(memory), _ptr(coeffs), len(inp))
        return out

    def filter_test(self, inp, memory, coeffs):
        inp = np.ascontiguousarray(inp, dtype=np.float32)
        out = np.zeros_like(inp)
        self.lib.wvo_filter_test(_ptr(inp), _ptr(out), _ptr(memory), _ptr(coeffs), len(inp))
        return out

    def directional_receiver(self, p7, spacing, sample_rate, ambient_density):
        p7 = np.ascontiguousarray(p7, dtype=np.float32)
        out = np.zeros((p7.shape[0], 4), dtype=np.float32)
        self.lib.wvo_directional_receiver(_ptr(p7), p7.shape[0], spacing, sample_rate,
                                          ambient_density, _ptr(out))
        return out


class ReferenceSetup:
    """The reference's mesh set-up kernels, host-compiled (oracle/_ref/libwvref_setup.so)."""

    def __init__(self):
        self.lib = C.CDLL(os.path.join(HERE, "_ref", "libwvref_setup.so"))
        self.lib.wvref_set_node_boundary_type.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]
        self.lib.wvref_set_node_boundary_type.restype = None

    @staticmethod
    def available():
        return os.path.exists(os.path.join(HERE, "_ref", "libwvref_setup.so"))

    def nodes_inside(self, dims, min_corner, spacing, voxel_index, aabb, side, triangles, vertices):
        """The reference's `set_node_inside` kernel: uint8 mask [nz, ny, nx]."""
        from wayverb_amd import mesh as M
        nx, ny, nz = dims
        nodes = np.zeros(nx * ny * nz, dtype=M.condensed_node_dtype)
        mc = np.ascontiguousarray(min_corner, dtype=np.float32)
        a0 = np.ascontiguousarray(aabb[0], dtype=np.float32)
        a1 = np.ascontiguousarray(aabb[1], dtype=np.float32)
        f = self.lib.wvref_set_node_inside
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_uint32, C.c_void_p, C.c_void_p]
        f(_ptr(nodes), nx, ny, nz, float(spacing), _ptr(mc), _ptr(voxel_index), _ptr(a0), _ptr(a1), side,
          _ptr(triangles), _ptr(vertices))
        return (nodes["boundary_type"] == 1).astype(np.uint8).reshape(nz, ny, nx)

    def set_node_boundary_type(self, inside_mask):
        from wayverb_amd import mesh as M
        nz, ny, nx = inside_mask.shape
        nodes = np.zeros(nx * ny * nz, dtype=M.condensed_node_dtype)
        nodes["boundary_type"] = np.where(inside_mask.reshape(-1), 1, 0)
        self.lib.wvref_set_node_boundary_type(_ptr(nodes), nx, ny, nz, 0.1)
        return nodes

    def boundary_coefficient_finder(self, nodes, dims, min_corner, spacing, triangles, vertices, counts):
        """The reference's boundary_coefficient_finder_{1,2,3}d kernels (serial, in index order) on
        first-numbering nodes: (out1 [n1,1], out2 [n2,2], out3 [n3,3])."""
        lib = C.CDLL(os.path.join(HERE, "_ref", "libwvref_bcf.so"))
        nx, ny, nz = dims
        mc = np.ascontiguousarray(min_corner, dtype=np.float32)
        o = [np.zeros((counts[d], d + 1), dtype=np.uint32) for d in range(3)]
        f = lib.wvref_boundary_coefficient_finder
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                      C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        f(_ptr(nodes), nx, ny, nz, float(spacing), _ptr(mc), _ptr(triangles), triangles.shape[0], _ptr(vertices),
          _ptr(o[0]), counts[0], _ptr(o[1]), counts[1], _ptr(o[2]), counts[2])
        return o


def reference_available():
    return all(os.path.exists(os.path.join(HERE, "_ref", "libwvref_%s.so" % t)) for t in ("f32", "f64"))


class Reference:
    """The reference's own kernel, host-compiled (oracle/build_ref.py)."""

    def __init__(self, tag):
        self.tag = tag
        self.dtype = np.float32 if tag == "f32" else np.float64
        self.lib = C.CDLL(os.path.join(HERE, "_ref", "libwvref_%s.so" % tag))
        self._step = getattr(self.lib, "wvref_step_" + tag)
        self._step.restype = None
        self._step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_int), C.c_int]
        self._ft2 = getattr(self.lib, "wvref_filter_test_2_" + tag)
        self._ft2.argtypes = [C.c_void_p] * 4 + [C.c_int]
        self._ft = getattr(self.lib, "wvref_filter_test_" + tag)
        self._ft.argtypes = [C.c_void_p] * 4 + [C.c_int]

    def step(self, previous, current, mesh, bd, threads=1):
        nx, ny, nz = mesh.dims
        assert previous.dtype == self.dtype and current.dtype == self.dtype
        flag = C.c_int(0)
        self._step(_ptr(previous), _ptr(current), _ptr(mesh.nodes), nx, ny, nz,
                   _ptr(bd[0]), _ptr(bd[1]), _ptr(bd[2]), _ptr(mesh.coefficients),
                   C.byref(flag), threads)
        return int(flag.value)

    def run(self, buf0, buf1, mesh, bd, source_kind, source_node, signal, n_steps, recv, threads=1):
        """The loop of waveguide.h:80-125 around the reference kernel (python driver)."""
        previous, current = buf0, buf1
        out = np.zeros((n_steps, len(recv)), dtype=self.dtype)
        for step in range(n_steps):
            if source_kind == 1:
                current[source_node] = self.dtype(np.float32(signal[step])) if self.tag == "f32" \
                    else self.dtype(signal[step])
            elif source_kind == 2:
                current[source_node] = current[source_node] + self.dtype(signal[step])
            flag = self.step(previous, current, mesh, bd, threads)
            if flag:
                return step, flag, out
            out[step] = current[np.asarray(recv, dtype=np.int64)]
            previous, current = current, previous
        return n_steps, 0, out

    def filter_test_2(self, inp, memory, coeffs):
        inp = np.ascontiguousarray(inp, dtype=np.float32)
        out = np.zeros_like(inp)
        self._ft2(_ptr(inp), _ptr(out), _ptr(memory), _ptr(coeffs), len(inp))
        return out

    def filter_test(self, inp, memory, coeffs):
        inp = np.ascontiguousarray(inp, dtype=np.float32)
        out = np.zeros_like(inp)
        self._ft(_ptr(inp), _ptr(out), _ptr(memory), _ptr(coeffs), len(inp))
        return out


class ReferenceOnDevice:
    """The reference's OpenCL program run by the machine's own OpenCL runtime (oracle/ref_cl_driver.cpp
    + the program text inside oracle/_ref/libwvref_cl.so): on the GPU box, the reference's kernel on
    the MI355X through ROCm's OpenCL.  TEST / BENCH-BASELINE INFRASTRUCTURE ONLY.

    Runs in a worker process (oracle/ref_cl_worker.py): ROCm's OpenCL runtime and the HIP runtime
    bundled with torch bring one HSA runtime each, and a process that initialises both loses its GPU."""

    PATH = os.path.join(HERE, "_ref", "libwvref_cl.so")
    WORKER = os.path.join(HERE, "ref_cl_worker.py")

    @classmethod
    def built(cls):
        return os.path.exists(cls.PATH)

    def _worker(self, *args, timeout=3600):
        import subprocess
        import sys
        env = dict(os.environ)
        p = subprocess.run([sys.executable, self.WORKER] + [str(a) for a in args], capture_output=True, text=True,
                           timeout=timeout, env=env)
        if p.returncode != 0:
            raise RuntimeError("reference on the OpenCL device: " + (p.stderr.strip() or p.stdout.strip())[-2000:])
        return p.stdout

    def device_name(self):
        """Name of the OpenCL GPU device, or None when there is none."""
        try:
            out = self._worker("name").strip()
        except RuntimeError:
            return None
        return out or None

    def run(self, previous, current, mesh, bd, source_kind, source_node, signal, n_steps, recv, contract_off=False):
        """waveguide::run on the OpenCL device.  previous / current / bd are updated in place (roles as
        after the last swap).  Returns (steps, flag, traces[steps, len(recv)], seconds of the step loop)."""
        import tempfile
        assert previous.dtype in (np.float32, np.float64) and current.dtype == previous.dtype
        with tempfile.TemporaryDirectory(prefix="wvrefcl_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
            fin, fout = os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")
            np.savez(fin, dims=np.array(mesh.dims), nodes=mesh.nodes, coefficients=mesh.coefficients, bd1=bd[0], bd2=bd[1],
                     bd3=bd[2], previous=previous, current=current, source_kind=source_kind, source_node=source_node,
                     signal=np.asarray(signal, dtype=np.float64), n_steps=n_steps, recv=np.asarray(recv, dtype=np.uint64),
                     contract_off=bool(contract_off))
            self._worker("run", fin, fout)
            out = np.load(fout)
            previous[...] = out["previous"]
            current[...] = out["current"]
            for dst, key in zip(bd, ("bd1", "bd2", "bd3")):
                dst[...] = out[key]
            return int(out["steps"]), int(out["flag"]), out["trace"], float(out["seconds"])

    def mesh_setup(self, dims, min_corner, spacing, voxel_index, aabb, side, triangles, vertices, contract_off=True):
        """set_node_inside + set_node_boundary_type (src/waveguide/src/mesh.cpp:75-111) on the OpenCL device:
        condensed_node[n] as (boundary_type int32, boundary_index uint32) pairs, boundary_index still 0."""
        import tempfile
        with tempfile.TemporaryDirectory(prefix="wvrefcl_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
            fin, fout = os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")
            np.savez(fin, dims=np.array(dims), min_corner=np.asarray(min_corner, dtype=np.float32), spacing=np.float32(spacing),
                     voxel_index=np.asarray(voxel_index, dtype=np.uint32), aabb_c0=np.asarray(aabb[0], dtype=np.float32),
                     aabb_c1=np.asarray(aabb[1], dtype=np.float32), side=side, triangles=np.asarray(triangles, dtype=np.uint32),
                     vertices=np.asarray(vertices, dtype=np.float32), contract_off=bool(contract_off))
            self._worker("setup", fin, fout)
            return np.load(fout)["nodes"].reshape(-1, 2)

    def boundary_coefficient_finder(self, dims, min_corner, spacing, nodes, counts, triangles, vertices, contract_off=True):
        """boundary_coefficient_finder_{1,2,3}d (src/waveguide/src/boundary_coefficient_finder.cpp:73-124) on the OpenCL
        device; `nodes` with the FIRST numbering of compute_boundary_index_data, `counts` the three array lengths.
        Entry 0 of the 1-D array is raced for by every inside node on a real device and means nothing."""
        import tempfile
        with tempfile.TemporaryDirectory(prefix="wvrefcl_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
            fin, fout = os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")
            np.savez(fin, dims=np.array(dims), min_corner=np.asarray(min_corner, dtype=np.float32), spacing=np.float32(spacing),
                     nodes=np.ascontiguousarray(nodes), counts=np.array(counts), triangles=np.asarray(triangles, dtype=np.uint32),
                     vertices=np.asarray(vertices, dtype=np.float32), contract_off=bool(contract_off))
            self._worker("bcf", fin, fout)
            out = np.load(fout)
            return [out["b1"], out["b2"], out["b3"]]

    def bench(self, n, steps, tag):
        """Gnode-updates/s of the reference's kernel on an n^3 box (built inside the worker): a dict."""
        import json
        return json.loads(self._worker("bench", n, steps, tag).strip().splitlines()[-1])
