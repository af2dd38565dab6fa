// dd_common.h -- shared host/device arithmetic for libdandd_hip (gfx950 only).
//
// The arithmetic follows the published algorithms Dashing uses behind the command lines
// DandD builds at /root/reference/lib/sketch_classes.py:312,358-365,370-372 (SURVEY.md
// Appendix A): Thomas Wang's 64-bit mix, HyperLogLog (idx = top p bits, rho from the
// remaining 64-p bits), byte-max union, Ertl's ML estimator.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define DD_HD __host__ __device__ __forceinline__
#define DD_D __device__ __forceinline__

namespace dd {

// Thomas Wang 64-bit integer mix.
DD_HD uint64_t wang64(uint64_t key) {
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

// 128-bit canonical k-mer (k in 33..64) -> 64-bit hash input.  Engine-defined extension
// (Dashing stops at k = 32, /root/reference/lib/huffman_dandd.py:109).
DD_HD uint64_t fold128(uint64_t hi, uint64_t lo) { return lo ^ (hi * 0x9E3779B97F4A7C15ull); }

// Byte-wise max of two words whose bytes are all < 128 (HLL registers are <= 61).
DD_HD uint32_t bmax4(uint32_t a, uint32_t b) {
    uint32_t t = (a | 0x80808080u) - b;            // bit 7 of each byte: a_byte >= b_byte
    uint32_t m = ((t >> 7) & 0x01010101u) * 0xFFu;  // 0xFF where a >= b
    return (a & m) | (b & ~m);
}

// Ertl 2017, Algorithm 8: ML estimate from the 64-bin register histogram.
// Built with -ffp-contract=off; relerr = 1e-2 / sqrt(m) is computed once on the host
// (mle_relerr) so the device copy never evaluates a square root.
DD_HD double ertl_mle(const uint32_t* c, int p, double relerr) {
    const int q = 64 - p;
    const uint64_t m = 1ull << p;
    if (c[q + 1] == m) return INFINITY;
    int kmin, kmax;
    for (kmin = 0; c[kmin] == 0; ++kmin) {}
    int kminp = kmin > 1 ? kmin : 1;
    for (kmax = q + 1; kmax && c[kmax] == 0; --kmax) {}
    int kmaxp = kmax < q ? kmax : q;
    double z = 0.0;
    for (int k = kmaxp; k >= kminp; --k) z = 0.5 * z + (double)c[k];
    z = ldexp(z, -kminp);
    double cprime = (double)c[q + 1];
    if (q >= 1) cprime += (double)c[kmaxp];
    double a = z + (double)c[0];
    double mprime = (double)(m - c[0]);
    double b = z + ldexp((double)c[q + 1], -q);
    double x = (b <= 1.5 * a) ? mprime / (0.5 * b + a) : (mprime / b) * log1p(b / a);
    double dx = x, gprev = 0.0;
    while (dx > x * relerr) {
        int kappam1;
        frexp(x, &kappam1);
        int sh = (kmaxp + 1 > kappam1 + 2) ? kmaxp + 1 : kappam1 + 2;
        double xp = ldexp(x, -sh);
        double xp2 = xp * xp;
        double h = xp - xp2 / 3.0 + (xp2 * xp2) * (1.0 / 45.0 - xp2 / 472.5);
        for (int k = kappam1; k >= kmaxp; --k) {
            double hp = 1.0 - h;
            h = (xp + h * hp) / (xp + hp);
            xp += xp;
        }
        double g = cprime * h;
        for (int k = kmaxp - 1; k >= kminp; --k) {
            double hp = 1.0 - h;
            h = (xp + h * hp) / (xp + hp);
            xp += xp;
            g += (double)c[k] * h;
        }
        g += x * a;
        if (gprev < g && g <= mprime)
            dx *= (g - mprime) / (gprev - g);
        else
            dx = 0.0;
        x += dx;
        gprev = g;
    }
    return x * (double)m;
}

inline double mle_relerr(int p) { return 1e-2 / sqrt((double)(1ull << p)); }

// splitmix64 finaliser: the counter-based generator of the synthetic FASTA.
DD_HD uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

}  // namespace dd
