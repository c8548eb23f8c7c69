// wave_reduce.hpp -- sixteen wave-wide sums for the price of (almost) one: the batched reduction of the matrix-core
// factorisation's panel (kernels_factor_fast.hip: panel_rows).
//
// Every lane brings x[0 .. 16); afterwards every lane holds  sum over the 64 lanes of x[lane >> 2]  -- value k lands in quad k.
// Six exchange stages, and in each of the first four the two halves of the exchange carry DIFFERENT values (a transposition as
// much as a reduction), so 16 values cost 8 + 4 + 2 + 1 + 1 + 1 additions instead of 16 x 6:
//   lanes l ^ 32   v_permlane32_swap (gfx950): x[j], x[j + 8] -> one register, lanes < 32 hold x[j]'s pair sums, lanes >= 32 x[j + 8]'s
//   16-lane rows   v_permlane16_swap (gfx950): registers j, j + 4 -> row r of the result holds value j + 4 r
//   l ^ 15         DPP row_mirror, the two registers chosen by bit 3 of the lane
//   l ^ 7          DPP row_half_mirror, chosen by bit 2
//   l ^ 1, l ^ 2   DPP quad_perm, nothing left to pack
// (the xor masks 32, 16, 15, 7, 1, 2 span all 64 lanes).  scripts/probes/probe_reduce16.hip pins the lane semantics on the GPU.
#pragma once
#include <hip/hip_runtime.h>

namespace socp {
namespace devsolver {

template <int CTRL>
__device__ __forceinline__ double dpp_move(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// a <- [a's lanes 0-31 | b's lanes 0-31],  b <- [a's lanes 32-63 | b's lanes 32-63]
__device__ __forceinline__ void swap_halves(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
// rows of 16 lanes: a <- [a.row0, b.row0, a.row2, b.row2],  b <- [a.row1, b.row1, a.row3, b.row3]
__device__ __forceinline__ void swap_rows(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}

// one wave-wide sum, the result in every lane: four DPP stages inside the rows of 16 lanes, then the four row sums by v_readlane
__device__ __forceinline__ double wave_sum(double x)
{
    x += dpp_move<0xB1>(x);                                                  // quad_perm [1, 0, 3, 2]
    x += dpp_move<0x4E>(x);                                                  // quad_perm [2, 3, 0, 1]
    x += dpp_move<0x141>(x);                                                 // row_half_mirror: l ^ 7
    x += dpp_move<0x140>(x);                                                 // row_mirror: l ^ 15
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(x), 16 * k), hi = __builtin_amdgcn_readlane(__double2hiint(x), 16 * k);
        r[k] = __hiloint2double(hi, lo);
    }
    return (r[0] + r[1]) + (r[2] + r[3]);
}

__device__ __forceinline__ double reduce16(double (&x)[16], int lane)
{
    double y[8], z[4], u[2];
#pragma unroll
    for (int j = 0; j < 8; j++) { swap_halves(x[j], x[j + 8]); y[j] = x[j] + x[j + 8]; }
#pragma unroll
    for (int i = 0; i < 4; i++) { swap_rows(y[i], y[i + 4]); z[i] = y[i] + y[i + 4]; }
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const double mine = b3 ? z[i + 2] : z[i], theirs = b3 ? z[i] : z[i + 2];
        u[i] = mine + dpp_move<0x140>(theirs);                              // row_mirror: lane l ^ 15
    }
    const double mine = b2 ? u[1] : u[0], theirs = b2 ? u[0] : u[1];
    double v = mine + dpp_move<0x141>(theirs);                              // row_half_mirror: lane l ^ 7
    v += dpp_move<0xB1>(v);                                                  // quad_perm [1, 0, 3, 2]
    v += dpp_move<0x4E>(v);                                                  // quad_perm [2, 3, 0, 1]
    return v;
}

}  // namespace devsolver
}  // namespace socp
