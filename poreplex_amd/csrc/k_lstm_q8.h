// k_lstm_q8.h -- device helpers shared by the two translation units of the int8 LSTM kernels
// (k_lstm_q8.hip: K2 + the host side; k_lstm_q8_demux.hip: K5a / K5b, built with another instruction
// scheduler -- Makefile): digit products on v_mfma_i32_16x16x64_i8, the level combine, digit planes in LDS.
#pragma once
#include "pxg_common.h"
#include "k_lstm_shared.h"

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4i mfma8(v4i a, v4i b, v4i c)
{
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
}

struct Q8Acc { v4i a0, a1, a2, a3; };      // the four significance levels of one gate tile

// one 64-wide k block: eight digit products into the four levels
template <bool FIRST>
__device__ __forceinline__ void q8_block(Q8Acc& A, v4i w2, v4i w1, v4i w0, const v4i (&h)[3])
{
    const v4i z = {0, 0, 0, 0};
    A.a0 = mfma8(w2, h[2], FIRST ? z : A.a0);
    A.a1 = mfma8(w2, h[1], FIRST ? z : A.a1);
    A.a2 = mfma8(w2, h[0], FIRST ? z : A.a2);
    A.a3 = mfma8(w1, h[0], FIRST ? z : A.a3);
    A.a1 = mfma8(w1, h[2], A.a1);
    A.a2 = mfma8(w1, h[1], A.a2);
    A.a3 = mfma8(w0, h[1], A.a3);
    A.a2 = mfma8(w0, h[2], A.a2);
}

// levels -> pre-activation in table units (two int32 words, two conversions, two fma)
__device__ __forceinline__ float q8_combine(int a0, int a1, int a2, int a3, float S, float start)
{
    const int V = (a0 << 8) + a1;
    const int U = (a2 << 8) + a3;
    const float t = __builtin_fmaf((float)V, 65536.0f, (float)U);
    return __builtin_fmaf(t, S, start);
}

// g * 2^(-p-14) of the four gate blocks (i, f, c, o) of one layer: every block has its own weight exponent p (the largest
// that keeps the block's biggest weight within three balanced digits), g = 16 or 32 = the table units (kernel argument:
// four scalar registers, no vector register)
struct Q8Scale { float g[4]; };
inline Q8Scale q8_scale(const float (&s)[4]) { return Q8Scale{{s[0], s[1], s[2], s[3]}}; }

template <int NT>
__device__ __forceinline__ void q8_combine_tiles(f32x4 (&u)[NT], const Q8Acc (&A)[NT], const Q8Scale& S,
                                                 const f32x4 (&start)[NT])
{
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            u[nt][r] = q8_combine(A[nt].a0[r], A[nt].a1[r], A[nt].a2[r], A[nt].a3[r], S.g[r], start[nt][r]);
}

// q + 0x808080 of q = rint(h * 2^22): its three low bytes, each XOR 0x80, are the balanced digits
__device__ __forceinline__ unsigned q8_biased(float h)
{
    // fma(h, 2^22, 1.5 * 2^23) = 0x4B400000 + q as bits (round to nearest even, |q| <= 2^22)
    return __float_as_uint(__builtin_fmaf(h, 4194304.0f, 12582912.0f)) - (0x4B400000u - 0x00808080u);
}

// the NT (3 or 4) cells of a lane -> one dword per digit plane (byte nt = digit of cell nt)
template <int NT>
__device__ __forceinline__ void q8_pack(const float (&h)[NT], unsigned (&plane)[3])
{
    unsigned q[4];
#pragma unroll
    for (int i = 0; i < NT; i++) q[i] = q8_biased(h[i]);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        // v_perm_b32: selector byte 0-3 = byte of the second operand, 4-7 = byte of the first, 0x0c = 0
        const unsigned lo = __builtin_amdgcn_perm(q[1], q[0], 0x0c0c0000u | ((4u + d) << 8) | (unsigned)d);
        unsigned v;
        if (NT == 3) {
            v = __builtin_amdgcn_perm(q[2], lo, 0x0c000100u | ((4u + d) << 16));
            plane[d] = v ^ 0x00808080u;
        } else {
            const unsigned hi = __builtin_amdgcn_perm(q[3], q[2], 0x0c0c0000u | ((4u + d) << 8) | (unsigned)d);
            v = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            plane[d] = v ^ 0x80808080u;
        }
    }
}

// digit planes of one hidden vector in LDS: [3 planes][4 k groups][16 reads][16 bytes] = 3 KB
#define Q8_PLANE 1024
#define Q8_HVEC (3 * Q8_PLANE)

__device__ __forceinline__ void q8_load_b(v4i (&b)[3], const unsigned char* hvec, int lane)
{
#pragma unroll
    for (int d = 0; d < 3; d++)
        b[d] = *reinterpret_cast<const v4i*>(hvec + d * Q8_PLANE + (((lane >> 4) * 16 + (lane & 15)) << 4));
}

__device__ __forceinline__ void q8_store_h(unsigned char* hvec, const unsigned (&plane)[3], int slice, int lane)
{
#pragma unroll
    for (int d = 0; d < 3; d++)
        *reinterpret_cast<unsigned*>(hvec + d * Q8_PLANE + ((slice * 16 + (lane & 15)) << 4) + ((lane >> 4) << 2)) = plane[d];
}

// h (float, exact) of unit `unit` (wave-major numbering, `UPW` units per wave) of read `rd`
template <int UPW>
__device__ __forceinline__ float q8_read_h(const unsigned char* hvec, int rd, int unit)
{
    const int w = unit / UPW, nt = (unit % UPW) >> 2, ul = unit & 3;
    const int off = ((w * 16 + rd) << 4) + (ul << 2) + nt;
    const int q = (int)(signed char)hvec[off] + 256 * (int)(signed char)hvec[Q8_PLANE + off] +
                  65536 * (int)(signed char)hvec[2 * Q8_PLANE + off];
    return (float)q * (1.0f / 4194304.0f);
}

