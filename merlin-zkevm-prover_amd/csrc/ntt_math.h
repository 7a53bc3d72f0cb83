// ntt_math.h -- in-register small DFTs over Goldilocks used by the Stockham NTT passes and the FRI fold.
//
// In Goldilocks 2 has multiplicative order 192 (2^96 = -1), and the reference's root table
// (w(4) = 4096 = 2^12, w(5) = 64, w(6) = 8) makes every twiddle of a 16-point DFT a
// power of two: w_64^j = 2^(3 j), w_16^j = 2^(12 j).
#pragma once
#include "gl_math.h"

namespace nttm {

// 2^e mod p for 0 <= e < 192, evaluated at compile time
constexpr u64 cpow2(int e)
{
    unsigned __int128 r = 1;
    for (int i = 0; i < e; i++) {
        r <<= 1;
        if (r >= (unsigned __int128)GL_P) r -= GL_P;
    }
    return (u64)r;
}

// w_64^j (forward) and w_64^-j (inverse), j < 32.  w_64 = 8 = 2^3; 2^-3j = 2^(192 - 3 j).
struct W64Table { u64 fwd[32], inv[32]; };
constexpr W64Table make_w64()
{
    W64Table t{};
    for (int j = 0; j < 32; j++) {
        t.fwd[j] = cpow2(3 * j);
        t.inv[j] = cpow2((192 - 3 * j) % 192);
    }
    return t;
}
static constexpr W64Table W64 = make_w64(); // built at compile time; indices below are constants after unrolling

template <bool INV>
MI_HD constexpr u64 w64(int j)
{
    return INV ? W64.inv[j] : W64.fwd[j];
}

template <int Q>
MI_HD constexpr int bitrev(int k)
{
    int r = 0;
    for (int i = 0; i < Q; i++) r |= ((k >> i) & 1) << (Q - 1 - i);
    return r;
}

// 2^Q-point DFT (Q <= 6) in registers, natural order in and out, canonical values in and out.
// DIF radix-2 layers; the closing bit-reversal is a compile-time register renaming.
template <int Q, bool INV>
MI_HD void dft_reg(u64 (&x)[1 << Q])
{
    constexpr int N = 1 << Q;
#pragma unroll
    for (int len = N; len >= 2; len >>= 1) {
        const int half = len >> 1;
#pragma unroll
        for (int s = 0; s < N; s += len) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                u64 u = x[s + j], v = x[s + j + half];
                x[s + j] = gl::add(u, v);
                u64 d = gl::sub(u, v);
                x[s + j + half] = (j == 0) ? d : gl::mul(d, w64<INV>(j * (64 / len)));
            }
        }
    }
    u64 y[N];
#pragma unroll
    for (int k = 0; k < N; k++) y[k] = x[bitrev<Q>(k)];
#pragma unroll
    for (int k = 0; k < N; k++) x[k] = y[k];
}

} // namespace nttm
