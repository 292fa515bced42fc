// device_common.hip.h -- shared device-side definitions for the gfx950 waveguide kernels.
//
// Written for CDNA4 only: wave64, DPP wave shifts, 16-byte per-lane global accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wv {

// ---- per-node class map: 2 bits per node, 4 nodes per byte along x ---------------------------
// Replaces the reference's 8-byte condensed_node in the streaming kernel
// (src/waveguide/include/waveguide/cl/structs.h:19-22; todo.md:11).
//   bit0 set  -> node takes the 7-point update (id_inside / id_reentrant, program.cpp:442-445)
//   value 2   -> boundary node: written by the boundary kernel, never by the streaming kernel
//   value 0   -> id_none: rewritten to 0 every step (program.cpp:485,529)
enum : uint32_t { CLS_NONE = 0, CLS_INSIDE = 1, CLS_BOUNDARY = 2, CLS_REENTRANT = 3 };

// error_code bits (cl/structs.h:8-15)
enum : int { FLAG_INF = 1, FLAG_NAN = 2, FLAG_OUTSIDE_RANGE = 4, FLAG_OUTSIDE_MESH = 8,
             FLAG_SUSPICIOUS = 16 };

constexpr uint32_t INVALID_NODE = 0xFFFFFFFFu;

// Class-map addressing.  One 32-bit word holds 4 rows x 4 nodes: byte (y & 3) of word
// [z][y >> 2][x >> 2] carries the 2-bit classes of nodes x&~3 .. x|3 of row y, so a wave fetches
// the classes of all the rows of its tile with ONE coalesced dword load.  cls_pitch = words per
// row group = pitch / 4.
__host__ __device__ inline int64_t cls_word_index(int x, int y, int z, int ny, int cls_pitch) {
    return ((int64_t)z * ((ny + 3) >> 2) + (y >> 2)) * cls_pitch + (x >> 2);
}
__host__ __device__ inline int64_t cls_byte_index(int x, int y, int z, int ny, int cls_pitch) {
    return cls_word_index(x, y, z, ny, cls_pitch) * 4 + (y & 3);
}

template <typename Real>
struct Vec16;
template <>
struct Vec16<double> {
    typedef double type __attribute__((ext_vector_type(2)));
    static constexpr int N = 2;
};
template <>
struct Vec16<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static constexpr int N = 4;
};

// ---- wave64 neighbour exchange through DPP (no LDS) -------------------------------------------
// lane i receives `v` of lane i-1; lane 0 keeps `edge`.
__device__ __forceinline__ float lane_from_below(float edge, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v),
                                                      0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
// lane i receives `v` of lane i+1; lane 63 keeps `edge`.
__device__ __forceinline__ float lane_from_above(float edge, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v),
                                                      0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ double lane_from_below(double edge, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_from_above(double edge, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// value of `v` in lane `src` (a compile-time constant), broadcast to the whole wave
__device__ __forceinline__ float read_lane(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double read_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// x / 3, correctly rounded, without the divide sequence.  The update `s = s / 3` (program.cpp:408,
// courant_sq... `/ 3`) is the only division of the 7-point update and a full IEEE division costs ~11
// instructions (div_scale x2, rcp, 5 fma, div_fmas, div_fixup) -- a quarter of the two-step pass's inner loop.
// Markstein's scheme for a constant divisor: q0 = RN(x * c) with c = RN(1/3); r = x - 3 q0 exactly (one
// FMA); q1 = RN(q0 + r c) IS the correctly rounded quotient for every finite x (the one divisor pattern
// for which the scheme can miss, an all-ones significand, is not 3's); v_div_fixup then supplies IEEE's
// answers for inf / nan / signed zero.  Checked against the hardware division on every one of the 2^32
// floats and on 2^34 doubles incl. all exponents and subnormals (tests/test_gpu_div3.py).
// The LOW RANGE, where Markstein's relative-error argument does not apply (quotient subnormal or in the lowest normal
// binade, |x| < 3 * 2^-1021): there x, q0 and q1 are all integer multiples of u = 2^-1074, say x = X u with
// X < 3 * 2^53.  q0 = Q u with Q = round(X c) (gradual underflow rounds to whole units), and X c = X/3 (1 - 2^-54)
// lies within X/3 * 2^-54 < 1/2 of X/3, so |Q - X/3| < 1 and the integer X - 3 Q is one of -2 .. 2: r is exact (an FMA
// rounds once, and a handful of units is representable).  q1 = round(Q + (X - 3 Q) c), again ONE rounding of the exact
// value by the FMA: the fraction added is 0, +-0.333.. or +-0.666.. (each shrunk by 2^-54, never near a half), so q1 =
// Q, Q or Q +- 1 -- in each case the integer nearest X/3 = Q + (X - 3 Q)/3, whose fractional part is 0, 1/3 or 2/3 and
// therefore never a tie.  That is the correctly rounded quotient.  (f64 denormals are never flushed on gfx9.)
// tests/hip/div3_check.hip runs the whole bottom of that range (every pattern below 2^32, both signs) and 2^32 more
// patterns under the exponent fields 0..3, with windows where x / 3 crosses a binade.
__device__ __forceinline__ double div3(double x) {
    constexpr double c = 0x1.5555555555555p-2;
    const double q0 = x * c;
    const double r = __builtin_fma(-3.0, q0, x);
    return __builtin_amdgcn_div_fixup(__builtin_fma(r, c, q0), 3.0, x);
}
__device__ __forceinline__ float div3(float x) {
    constexpr float c = 0x1.555556p-2f;
    const float q0 = x * c;
    const float r = __builtin_fmaf(-3.0f, q0, x);
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(r, c, q0), 3.0f, x);
}

// neither inf nor nan: one v_cmp_class
__device__ __forceinline__ bool is_finite(float v) { return __builtin_amdgcn_classf(v, 0x1F8); }
__device__ __forceinline__ bool is_finite(double v) { return __builtin_amdgcn_class(v, 0x1F8); }

template <typename Real>
__device__ __forceinline__ int bad_bits(Real v) {
    return (isinf(v) ? FLAG_INF : 0) | (isnan(v) ? FLAG_NAN : 0);
}

// ---- kernel argument blocks -----------------------------------------------------------------
template <typename Real>
struct BoundaryArgs {
    const Real* prev;        // field at t-1: this node's own old value
    Real* next;              // where the node's new value goes (= prev when the update is in place)
    const Real* cur;
    int* flag;
    const uint32_t* bnode;   // [n_entries] local node index, INVALID_NODE = unused slot
    const uint8_t* btype;    // [n_entries] direction bits (boundary_type >> 1)
    double* fmem;            // [6][n_slots] filter memories, structure-of-arrays
    const uint32_t* cidx;    // [n_slots] coefficient (surface) index per filter
    const double* coeffs;    // [n_coeffs][14] = {b[7], a[7]}
    uint32_t n_coeffs;
    uint32_t n1, n2, n3;     // entries per dimensionality; entry order: all 1D, all 2D, all 3D
    uint32_t n_slots;        // n1 + 2 n2 + 3 n3
    int nx, ny, nz;
    int pitch;               // stored row length (see stream_kernels.hip.h)
    int z_begin, z_end;
    Real courant, courant_sq;
    const uint32_t* order;   // standalone kernel: entry ids to process (null = all, in list order)
    uint32_t n_order;
    // Second boundary launch of a two-step pass only (prev / cur / next = fields t / t+1 / t+2): a 1-D entry
    // also finishes the t+2 value of the inside node it faces -- a node the march had to leave open
    // because this boundary node's t+1 value did not exist yet -- when it lies in planes [fix_z0, fix_z1).
    // Its cache lines are the ones the entry touches anyway (boundary_kernel<.., FIX = true>; pair_kernels.hip.h,
    // pair_map_kernel).
    int fix_z0, fix_z1;
    // Two-step passes, x-facing walls on compact copies (boundary_kernels.hip.h, xwall_node): the first xw_n entries;
    // xw_pad = xw_n rounded up to whole workgroups; 0 = every entry gathers from the fields.
    uint32_t xw_n, xw_pad;
    const uint32_t* xw_nbr;                  // [4][xw_n]
    Real *xw_a, *xw_b, *xw_f, *xw_f1, *xw_g;  // [xw_n] each
    // ... and in three-step passes (xwall3_node): two more generations, the wall node's own value at a third time level and the faced
    // node at t+2
    Real *xw_o2, *xw_f2;
    // ... and for the node BEHIND the faced one, which such an entry finishes at the third level too where xw_gok says it may (a plain
    // node among plain nodes: xwall_cover_kernel): that node and the one behind it at t+2, captured by level 2
    Real *xw_g2, *xw_h2;
    const uint8_t* xw_gok;
};

template <typename Real>
struct StreamArgs {
    const Real* prev;    // previous field
    Real* next;          // where the next field goes: = prev (in place) except for the face planes of a
                         // slab's two-step pass, which go to another field
    const Real* cur;     // current field (read only)
    const uint8_t* cls;  // class map (see cls_word_index), cls_pitch = pitch/4 words per row group
    int* flag;           // error_code word of this step
    int nx, ny, nz;
    int pitch;           // elements per stored row: nx rounded up to 64 lanes x 16 B
    int cls_pitch;
    int z_begin, z_end;  // planes this engine updates (ghost planes excluded)
    // plane sweep only: the launch covers z_end - z_begin planes, of which those from z_skip_from on lie z_skip planes
    // further up (two plane ranges in one launch: a slab's two faces).  z_skip = 0: one range.
    int z_skip_from, z_skip;
    int zc;              // planes marched by one workgroup
    int tiles_x, tiles_y, chunks_z;
    int total_tiles, tiles_per_xcd;
    // plane-sweep kernel only: rows per XCD stripe, tiles per stripe-plane, passes over z
    int stripe_rows, tiles_y_stripe, passes;
    // optional work list (rooms that leave part of the mesh outside): workgroup j of XCD k takes
    // tile_list[list_start[k] + j] = stripe << 48 | wave mask << 40 | z << 20 | tile-in-stripe-plane
    const uint64_t* tile_list;
    uint32_t list_start[9];
};

template <typename Real>
struct PrePostArgs {
    Real* cur;
    const double* signal;    // device copy of the source signal
    uint64_t signal_pos;     // sample index for this step (relative to *signal_base when that is set)
    const uint64_t* signal_base;  // device scalar: lets a captured batch of steps be replayed further along the signal
    uint64_t source_node;
    int source_kind;         // 0 none, 1 hard, 2 soft
    const uint64_t* recv;    // [n_recv]
    Real* recv_out;          // row of this step: [n_recv]
    uint32_t n_recv;
    int* flag;               // this step's error_code word, reset here (null: leave it alone) ...
    int* flag2;              // (two-step pass) the next step's word, reset with it; null otherwise
    int flag_init;           // ... to the mesh-static bits (0 for a well-formed mesh)
    int fused;               // non-zero: boundary_kernel's last workgroup does this work
    // Two-step pass, riding with the source / receiver work of step t+1: the (few) nodes of the fix-up list
    // -- t+2 of the source node's neighbours, from the t+1 field with the sample just put in.  Only when
    // no listed node has a boundary node for a neighbour (engine.hip, pair_list_early_ok_).
    const uint32_t* fix_nodes;
    uint32_t fix_n;
    const Real* fix_cur;     // field t (the listed node's own old value)
    Real* fix_out2;          // field t+2
    int* fix_flag;           // error_code word of step t+1 -> t+2
    int nx, ny, nz, pitch;
};

// One-launch steps (plane_kernels.hip.h, whole_step_kernel): the NEXT step's source sample and receiver samples are served by the
// sweep workgroup that produces the node's value, from its registers, before the value is stored.
struct StepDuty {
    uint64_t node;   // stored index (an inside / re-entrant node: a sweep tile owns it)
    uint32_t block;  // the sweep workgroup (arithmetic tile mapping, no work list) whose tile holds it
    uint32_t col;    // receiver: column of the step's row
    uint32_t kind;   // 0 receiver, 1 hard source, 2 soft source (a source's duty comes first in the list)
    uint32_t pad_;
};
template <typename Real>
struct StepDuties {
    const StepDuty* list;  // [n], n <= 64
    uint32_t n;
    const double* signal;        // the source signal (device)
    uint64_t signal_pos;         // sample of the NEXT step (relative to *signal_base when that is set)
    const uint64_t* signal_base;
    Real* recv_out;              // row of the next step: [n_recv]
    const uint64_t* recv;        // [n_recv] stored indices (~0: not recorded, its column gets 0)
    uint32_t n_recv;
    int* next_flag;              // the next step's error_code word, reset here (null: leave it alone)
    int flag_init;
};

}  // namespace wv
