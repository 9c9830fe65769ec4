// Shared host/device helpers of libbinf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/binf_hip.h"

// numpy rounds every multiply and add separately; EXACT-mode kernels must not
// contract a*b+c.  (The build also passes -ffp-contract=off.)
#pragma clang fp contract(off)

namespace binf {

// ---- error text (thread-local) -------------------------------------------
void set_error(const char *fmt, ...);
int32_t fail(int32_t code, const char *fmt, ...);
int32_t hip_fail(hipError_t e, const char *what);

#define BINF_HIP_CHECK(expr)                                   \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return binf::hip_fail(e_, #expr); \
    } while (0)

// ---- buffer aliasing (host) ------------------------------------------------
// true if [a, a + a_elems) and [b, b + b_elems) (doubles) share a byte
inline bool overlap_f64(const void *a, int64_t a_elems, const void *b, int64_t b_elems)
{
    if (!a || !b || a_elems <= 0 || b_elems <= 0) return false;
    const char *pa = (const char *)a, *pb = (const char *)b;
    return pa < pb + b_elems * (int64_t)sizeof(double) && pb < pa + a_elems * (int64_t)sizeof(double);
}

// ---- numpy pairwise-summation geometry -------------------------------------
// np.add.reduce on a contiguous f64 vector: blocks of <=128 elements ("leaves")
// are summed with 8 strided accumulators, blocks are joined by a binary tree
// that splits n at n/2 rounded down to a multiple of 8.
constexpr int PW_BLOCK = 128;

struct Leaf {
    int32_t off;        // first element of the leaf
    int32_t len;        // number of elements (<=128)
    int32_t depth;      // depth of the leaf in the tree (root = 0)
    int32_t canonical;  // 1 if `path` is the lowest path that reaches it
};

inline int32_t pairwise_tree_height(int64_t n)  // host only
{
    if (n <= PW_BLOCK) return 0;
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    int32_t a = pairwise_tree_height(n2);
    int32_t b = pairwise_tree_height(n - n2);
    return 1 + (a > b ? a : b);
}

// Walk from the root along `path` (bit H-1 first; 0 = left half).  A leaf
// that sits above depth H is reached by every path below it; those paths
// recompute the same leaf ("redundant groups"), which keeps the cross-leaf
// butterfly uniform.
__host__ __device__ inline Leaf pairwise_leaf(int32_t n, int32_t H, int32_t path)
{
    Leaf L;
    int32_t off = 0, depth = 0;
    for (int32_t d = 0; d < H; ++d) {
        if (n <= PW_BLOCK) break;
        int32_t n2 = n / 2;
        n2 -= n2 % 8;
        if ((path >> (H - 1 - d)) & 1) {
            off += n2;
            n -= n2;
        } else {
            n = n2;
        }
        ++depth;
    }
    L.off = off;
    L.len = n;
    L.depth = depth;
    L.canonical = ((path & ((1 << (H - depth)) - 1)) == 0) ? 1 : 0;
    return L;
}

// ---- wave64 cross-lane moves for doubles ----------------------------------
__device__ inline double shfl_xor_f64(double v, int mask)
{
    return __shfl_xor(v, mask, 64);
}
__device__ inline double shfl_f64(double v, int src)
{
    return __shfl(v, src, 64);
}

// The same moves through the DPP path of the VALU (a few cycles; __shfl is a
// ds_bpermute pair through the LDS crossbar, ~100 cycles of latency each way).
template <int CTRL>
__device__ inline double dpp_f64(double v)
{
    // every control used below reads a valid lane for every lane (all rows and banks
    // enabled), so no lane keeps an old value: bound_ctrl lets the move write a fresh
    // register instead of copying the source first (the copy sat on the critical path)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double xor1_f64(double v) { return dpp_f64<0xB1>(v); }    // quad_perm [1,0,3,2]
__device__ inline double xor2_f64(double v) { return dpp_f64<0x4E>(v); }    // quad_perm [2,3,0,1]
__device__ inline double xor8_f64(double v) { return dpp_f64<0x128>(v); }   // row_ror:8 (rows of 16)
// lane i <- lane 7 - i of its aligned group of 8 (row_half_mirror): the partner sits in
// the OTHER quad, which is all the third level of an 8-lane sum tree needs once the four
// lanes of a quad hold the same value -- then the same bits as an xor-4 exchange
__device__ inline double other_quad_f64(double v) { return dpp_f64<0x141>(v); }

// xor-16 / xor-32 partners through the gfx950 lane-swap instructions (VALU; the
// ds_bpermute pair they replace waits ~100 cycles for the LDS crossbar):
//   v_permlane16_swap vdst, src: odd rows (of 16 lanes) of vdst <-> even rows of src
//   v_permlane32_swap vdst, src: lanes 32..63 of vdst <-> lanes 0..31 of src
// with vdst = src = v, one result holds the even / lower copies and the other the
// odd / upper ones; a lane picks the one that carries its partner's value.
__device__ inline double xor16_f64(double v, int lane)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const bool odd = (lane >> 4) & 1;
    return __hiloint2double(odd ? (int)b[0] : (int)b[1], odd ? (int)a[0] : (int)a[1]);
}
__device__ inline double xor32_f64(double v, int lane)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const bool up = lane >= 32;
    return __hiloint2double(up ? (int)b[0] : (int)b[1], up ? (int)a[0] : (int)a[1]);
}
// partner lane ^ (8 << l), l = 0, 1, 2 (the leaf-tree levels inside a wave)
__device__ inline double xor_level_f64(double v, int l, int lane)
{
    return l == 0 ? xor8_f64(v) : (l == 1 ? xor16_f64(v, lane) : xor32_f64(v, lane));
}

// all-reduce sum over aligned groups of 8 lanes in the order
// ((v0+v1)+(v2+v3))+((v4+v5)+(v6+v7)); every lane ends with the same bits
__device__ inline double sum8_f64(double r)
{
    r = r + xor1_f64(r);
    r = r + xor2_f64(r);
    return r + other_quad_f64(r);
}

}  // namespace binf
