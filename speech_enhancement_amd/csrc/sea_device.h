/*
 * sea_device.h -- device-side building blocks shared by the gfx950 kernels.
 *
 * Everything here is written for ONE 64-lane wavefront working on one frame / one utterance with
 * its scratch in LDS.  The arithmetic reproduces the reference C expression by expression
 * (compile with -ffp-contract=off: no FMA may be formed from a*b+c), while the distribution of the
 * work over lanes is this engine's own.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sea_tables.h"

namespace sea {

constexpr int kLanes = 64;

/* LDS traffic of ONE wave is executed in program order by the hardware (DS instructions of a wave
 * issue and return in order), so lanes of the same wave can exchange data through LDS without any
 * hardware wait; what has to be prevented is the compiler moving a lane's read above another
 * lane's write.  A wavefront-scope release/acquire fence pair around the wave barrier pseudo-op
 * does exactly that and emits no instruction, so waves of one workgroup can run different code
 * (the pipelined kernel) without meeting at an s_barrier. */
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* per-lane constants of the FFT schedule (sea_fft_tables), loaded once, kept in VGPRs */
/* per-lane constants of the FFT schedule (sea_fft_tables), loaded once, kept in VGPRs */
struct FftRegs {
    unsigned flags;
    unsigned item[SEA_FFT_LSTAGES];
    float tw[SEA_FFT_LSTAGES][4];
};

__device__ __forceinline__ void load_fft_regs(FftRegs &R, const sea_fft_tables *t, int lane)
{
    R.flags = t->fftFlags[lane];
#pragma unroll
    for (int s = 0; s < SEA_FFT_LSTAGES; ++s) {
        R.item[s] = t->fftItem[s][lane];
#pragma unroll
        for (int k = 0; k < 4; ++k) R.tw[s][k] = t->fftTw[s][k][lane];
    }
}

/* One split-radix level n2 = 8<<S: every lane executes at most one butterfly (its work item).
 * Arithmetic: etsi/cpp/rfft.c:110-113 (plain), :120-125 (pi/4), :145-174 (twiddled). */
template <int S>
__device__ __forceinline__ void fft_level(float *work, const FftRegs &R)
{
    constexpr int n4 = 2 << S;
    const unsigned it = R.item[S];
    const unsigned kind = it >> 16;
    const int a = (int)(it & 255u);
    const int b = (int)((it >> 8) & 255u); /* == a for plain / pi4 items, 0 for idle lanes */
    /* All operands are fetched before the (divergent) arithmetic so that the three butterfly kinds
     * share ONE LDS round trip instead of paying one each. */
    const float x1 = work[a], x2 = work[a + n4], x3 = work[a + 2 * n4], x4 = work[a + 3 * n4];
    const float x5 = work[b], x6 = work[b + n4], x7 = work[b + 2 * n4], x8 = work[b + 3 * n4];
    float o1 = x1, o2 = x2, o3 = x3, o4 = x4, o5 = x5, o6 = x6, o7 = x7, o8 = x8;
    if (kind == SEA_BF_TWIDDLE) {
        const float cc1 = R.tw[S][0], ss1 = R.tw[S][1], cc3 = R.tw[S][2], ss3 = R.tw[S][3];
        float t1 = x3 * cc1 + x7 * ss1;
        float t2 = x7 * cc1 - x3 * ss1;
        float t3 = x4 * cc3 + x8 * ss3;
        float t4 = x8 * cc3 - x4 * ss3;
        const float t5 = t1 + t3, t6 = t2 + t4;
        t3 = t1 - t3;
        t4 = t2 - t4;
        o3 = t6 - x6;  /* x[i3] */
        o8 = x6 + t6;  /* x[i8] */
        o7 = -x2 - t3; /* x[i7] */
        o4 = x2 - t3;  /* x[i4] */
        o6 = x1 - t5;  /* x[i6] */
        o1 = x1 + t5;  /* x[i1] */
        o5 = x5 - t4;  /* x[i5] */
        o2 = x5 + t4;  /* x[i2] */
    } else if (kind == SEA_BF_PLAIN) {
        const float t1 = x4 + x3;
        o4 = x4 - x3;
        o3 = x1 - t1;
        o1 = x1 + t1;
    } else if (kind == SEA_BF_PI4) {
        /* Reference: float sum, DOUBLE division by M_SQRT2, rounded back to float (rfft.c:120-121).
         * (float)((double)s * (1/sqrt2)) gives the same float for EVERY float s: both doubles are
         * within 2^-52 (relative) of s/sqrt2, and s/sqrt2 can never be that close to a float
         * rounding boundary, because |sqrt2*m - (2k+1)| > 1/(2.83 m) for integers m < 2^24 keeps it
         * 2^-50.6 away.  tests/test_gpu_parity.py checks all 2^32 floats on the device. */
        const float t1 = (float)((double)(x3 + x4) * 0.70710678118654752440);
        const float t2 = (float)((double)(x3 - x4) * 0.70710678118654752440);
        o4 = x2 - t1;
        o3 = -x2 - t1;
        o2 = x1 - t2;
        o1 = x1 + t2;
    }
    if (kind != SEA_BF_NONE) {
        work[a] = o1;
        work[a + n4] = o2;
        work[a + 2 * n4] = o3;
        work[a + 3 * n4] = o4;
    }
    if (kind == SEA_BF_TWIDDLE) {
        work[b] = o5;
        work[b + n4] = o6;
        work[b + 2 * n4] = o7;
        work[b + 3 * n4] = o8;
    }
}

/* The FFT work area in LDS is XOR-swizzled per 32-word block (sea_tables.h::sea_fft_swizzle): with the
 * natural layout the lanes of a level hit 2..8 banks.  Butterfly operands, head stores and PSD reads
 * take their (byte) addresses from tables; the few other readers call fft_swz(). */
__device__ __forceinline__ unsigned fft_swz(unsigned i) /* element index -> BYTE offset in the work area */
{
    constexpr unsigned long long kSwz = 0ull | (6ull << 5) | (29ull << 10) | (15ull << 15) | (18ull << 20) | (20ull << 25) |
                                        (9ull << 30) | (27ull << 35); /* {0, 6, 29, 15, 18, 20, 9, 27} */
    return (i ^ (unsigned)((kSwz >> (5u * ((i >> 5) & 7u))) & 31ull)) * 4u;
}
__device__ __forceinline__ float &fft_at(float *work, unsigned off)
{
    return *reinterpret_cast<float *>(reinterpret_cast<char *>(work) + off);
}
__device__ __forceinline__ const float &fft_at(const float *work, unsigned off)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(work) + off);
}

struct FftRegsSwz {
    unsigned flags;
    unsigned kind[SEA_FFT_LSTAGES];
    unsigned addr[SEA_FFT_LSTAGES][4]; /* byte offsets of the eight operands, two per word (sea_tables.h) */
    float tw[SEA_FFT_LSTAGES][4];
    unsigned head[2], psd[2], nyq;     /* where this lane stores its head values / finds its PSD inputs */
};

__device__ __forceinline__ void load_fft_regs(FftRegsSwz &R, const sea_fft_tables *t, int lane)
{
    R.flags = t->fftFlags[lane];
#pragma unroll
    for (int s = 0; s < SEA_FFT_LSTAGES; ++s) {
        R.kind[s] = t->fftItem[s][lane] >> 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            R.addr[s][k] = t->fftAddr[s][k][lane];
            R.tw[s][k] = t->fftTw[s][k][lane];
        }
    }
    R.head[0] = t->fft2Head[0][lane];
    R.head[1] = t->fft2Head[1][lane];
    R.psd[0] = t->fft2Psd[0][lane];
    R.psd[1] = t->fft2Psd[1][lane];
    R.nyq = t->fft2Nyq;
}

/* The same level on the swizzled work area (table-driven addresses).
 * Arithmetic: etsi/cpp/rfft.c:110-113 (plain), :120-125 (pi/4), :145-174 (twiddled). */
template <int S>
__device__ __forceinline__ void fft_level(float *work, const FftRegsSwz &R)
{
    const unsigned kind = R.kind[S];
    const unsigned a01 = R.addr[S][0], a23 = R.addr[S][1], b01 = R.addr[S][2], b23 = R.addr[S][3];
    /* All operands are fetched before the (divergent) arithmetic so that the three butterfly kinds
     * share ONE LDS round trip instead of paying one each. */
    const float x1 = fft_at(work, a01 & 0xffffu), x2 = fft_at(work, a01 >> 16);
    const float x3 = fft_at(work, a23 & 0xffffu), x4 = fft_at(work, a23 >> 16);
    const float x5 = fft_at(work, b01 & 0xffffu), x6 = fft_at(work, b01 >> 16);
    const float x7 = fft_at(work, b23 & 0xffffu), x8 = fft_at(work, b23 >> 16);
    float o1 = x1, o2 = x2, o3 = x3, o4 = x4, o5 = x5, o6 = x6, o7 = x7, o8 = x8;
    if (kind == SEA_BF_TWIDDLE) {
        const float cc1 = R.tw[S][0], ss1 = R.tw[S][1], cc3 = R.tw[S][2], ss3 = R.tw[S][3];
        float t1 = x3 * cc1 + x7 * ss1;
        float t2 = x7 * cc1 - x3 * ss1;
        float t3 = x4 * cc3 + x8 * ss3;
        float t4 = x8 * cc3 - x4 * ss3;
        const float t5 = t1 + t3, t6 = t2 + t4;
        t3 = t1 - t3;
        t4 = t2 - t4;
        o3 = t6 - x6;  /* x[i3] */
        o8 = x6 + t6;  /* x[i8] */
        o7 = -x2 - t3; /* x[i7] */
        o4 = x2 - t3;  /* x[i4] */
        o6 = x1 - t5;  /* x[i6] */
        o1 = x1 + t5;  /* x[i1] */
        o5 = x5 - t4;  /* x[i5] */
        o2 = x5 + t4;  /* x[i2] */
    } else if (kind == SEA_BF_PLAIN) {
        const float t1 = x4 + x3;
        o4 = x4 - x3;
        o3 = x1 - t1;
        o1 = x1 + t1;
    } else if (kind == SEA_BF_PI4) {
        /* Reference: float sum, DOUBLE division by M_SQRT2, rounded back to float (rfft.c:120-121).
         * (float)((double)s * (1/sqrt2)) gives the same float for EVERY float s: both doubles are
         * within 2^-52 (relative) of s/sqrt2, and s/sqrt2 can never be that close to a float
         * rounding boundary, because |sqrt2*m - (2k+1)| > 1/(2.83 m) for integers m < 2^24 keeps it
         * 2^-50.6 away.  tests/test_gpu_parity.py checks all 2^32 floats on the device. */
        const float t1 = (float)((double)(x3 + x4) * 0.70710678118654752440);
        const float t2 = (float)((double)(x3 - x4) * 0.70710678118654752440);
        o4 = x2 - t1;
        o3 = -x2 - t1;
        o2 = x1 - t2;
        o1 = x1 + t2;
    }
    if (kind != SEA_BF_NONE) {
        fft_at(work, a01 & 0xffffu) = o1;
        fft_at(work, a01 >> 16) = o2;
        fft_at(work, a23 & 0xffffu) = o3;
        fft_at(work, a23 >> 16) = o4;
    }
    if (kind == SEA_BF_TWIDDLE) {
        fft_at(work, b01 & 0xffffu) = o5;
        fft_at(work, b01 >> 16) = o6;
        fft_at(work, b23 & 0xffffu) = o7;
        fft_at(work, b23 >> 16) = o8;
    }
}

/* exhaustive check kernel for the pi/4 identity above: counts floats s for which
 * (float)((double)s * inv_sqrt2) != (float)((double)s / sqrt2) */
__device__ __forceinline__ bool pi4_identity_holds(float s)
{
    const float fast = (float)((double)s * 0.70710678118654752440);
    const float ref = (float)((double)s / 1.41421356237309504880);
    return __float_as_uint(fast) == __float_as_uint(ref) || (fast != fast && ref != ref);
}

/* 256-point real split-radix FFT of one frame held 4 elements per lane: lane l passes elements
 * l, l+64, l+128, l+192 (already windowed / zero padded).  Result is left in work[0..255] in the
 * reference's order Re(0..128), Im(127..1) (etsi/cpp/rfft.c:27-29).  Ends with a wave_sync(). */
__device__ __forceinline__ void rfft256(float e0, float e1, float e2, float e3, float *work,
                                        const FftRegs &R, int lane)
{
    /* bit reversal: positions 4r..4r+3 (r = bitrev6(lane)) take elements l, l+128, l+64, l+192 */
    float g0 = e0, g1 = e2, g2 = e1, g3 = e3;
    {   /* length-two butterflies (rfft.c:82-96) */
        const float s01 = g0 + g1, d01 = g0 - g1, s23 = g2 + g3, d23 = g2 - g3;
        const bool f0 = (R.flags & 1u) != 0, f1 = (R.flags & 2u) != 0;
        g0 = f0 ? s01 : g0;
        g1 = f0 ? d01 : g1;
        g2 = f1 ? s23 : g2;
        g3 = f1 ? d23 : g3;
    }
    {   /* n2 = 4 level: plain butterfly only (rfft.c:110-113) */
        const float t1 = g3 + g2;
        const float n3 = g3 - g2, n2 = g0 - t1, n0 = g0 + t1;
        const bool f = (R.flags & 4u) != 0;
        g3 = f ? n3 : g3;
        g2 = f ? n2 : g2;
        g0 = f ? n0 : g0;
    }
    const int r = (int)(__brev((unsigned)lane) >> 26);
    *reinterpret_cast<float4 *>(work + 4 * r) = make_float4(g0, g1, g2, g3);
    wave_sync();
#ifndef SEA_ABLATE_FFT
    fft_level<0>(work, R);
    wave_sync();
    fft_level<1>(work, R);
    wave_sync();
    fft_level<2>(work, R);
    wave_sync();
    fft_level<3>(work, R);
    wave_sync();
    fft_level<4>(work, R);
    wave_sync();
    fft_level<5>(work, R);
    wave_sync();
#endif
}

/* Swizzled form of rfft256: 256-point real split-radix FFT of one frame held 4 elements per lane: lane l passes elements
 * l, l+64, l+128, l+192 (already windowed / zero padded).  Result: element i of the reference's order
 * Re(0..128), Im(127..1) (etsi/cpp/rfft.c:27-29) sits at byte offset fft_swz(i) of work.  Ends with a
 * wave_sync(). */
__device__ __forceinline__ void rfft256(float e0, float e1, float e2, float e3, float *work,
                                        const FftRegsSwz &R, int lane)
{
    /* bit reversal: positions 4r..4r+3 (r = bitrev6(lane)) take elements l, l+128, l+64, l+192 */
    float g0 = e0, g1 = e2, g2 = e1, g3 = e3;
    {   /* length-two butterflies (rfft.c:82-96) */
        const float s01 = g0 + g1, d01 = g0 - g1, s23 = g2 + g3, d23 = g2 - g3;
        const bool f0 = (R.flags & 1u) != 0, f1 = (R.flags & 2u) != 0;
        g0 = f0 ? s01 : g0;
        g1 = f0 ? d01 : g1;
        g2 = f1 ? s23 : g2;
        g3 = f1 ? d23 : g3;
    }
    {   /* n2 = 4 level: plain butterfly only (rfft.c:110-113) */
        const float t1 = g3 + g2;
        const float n3 = g3 - g2, n2 = g0 - t1, n0 = g0 + t1;
        const bool f = (R.flags & 4u) != 0;
        g3 = f ? n3 : g3;
        g2 = f ? n2 : g2;
        g0 = f ? n0 : g0;
    }
    fft_at(work, R.head[0] & 0xffffu) = g0;
    fft_at(work, R.head[0] >> 16) = g1;
    fft_at(work, R.head[1] & 0xffffu) = g2;
    fft_at(work, R.head[1] >> 16) = g3;
    wave_sync();
#ifndef SEA_ABLATE_FFT
    fft_level<0>(work, R);
    wave_sync();
    fft_level<1>(work, R);
    wave_sync();
    fft_level<2>(work, R);
    wave_sync();
    fft_level<3>(work, R);
    wave_sync();
    fft_level<4>(work, R);
    wave_sync();
    fft_level<5>(work, R);
    wave_sync();
#endif
}

/* ---- two independent 256-point transforms side by side in one wave (lanes 0..31 / 32..63) ----
 * Same butterflies, same arithmetic; the plain and pi/4 butterflies of a block share one work item
 * (SEA_BF_PAIR) so that no level needs more than 32 lanes per transform.  work holds the two
 * frames back to back: transform A in work[0..255], B in work[256..511]. */
/* ADDR_LDS selects where the 24 address words of a lane live: in VGPRs (fastest transform: the kernel is
 * then bound by the longest utterance's chain of frames, which is what matters up to four workgroups
 * per CU) or in LDS (one ds_read_b128 per level, 30 VGPRs fewer: six workgroups per CU, which is what
 * matters for large batches). */
#ifndef SEA_FFT_PACKED
#define SEA_FFT_PACKED 1
#endif

#ifndef SEA_FFT_HEAD16
#define SEA_FFT_HEAD16 1 /* the n2 = 16 level of the dual transform on registers (rfft256_head16): one LDS round trip less per transform.
                          * Only in the table-in-LDS forms (ADDR_LDS: the large-batch NoiseSup kernels, where the LDS array is the busier
                          * pipe: configs[4] shard 477 -> 481 M frames/s); everywhere else the ~25 extra vector instructions cost what the
                          * sixteen LDS instructions save (configs[1] 1.966 -> 1.981 ms, CompCeps unchanged) */
#endif
struct Fft2Regs {
    unsigned kind[SEA_FFT_LSTAGES];
    unsigned addr[SEA_FFT_LSTAGES][4]; /* byte offsets of the eight operands, two per word, already moved
                                        * into the second work area for lanes >= 32 (sea_tables.h) */
    const uint4 *addrLds;              /* ADDR_LDS: the same, [level][64 lanes] in LDS, this lane's column */
    float tw[SEA_FFT_LSTAGES][4];
    unsigned headA[2], psdA[2];        /* frame A: where this lane stores its head values / finds its PSD inputs */
    unsigned nyq;
    unsigned head8Flags, head8[4];     /* the eight-positions-per-lane start (sea_tables.h fft8*), this lane's transform */
    unsigned head16[4];                /* SEA_FFT_HEAD16: where the lane's results of the register-resident n2 = 16 level go (fft16*) */
    float tw16[4];                     /* the one twiddle set of that level (j = 1) */
};


/* addrLds (ADDR_LDS only): SEA_FFT_LSTAGES * 64 uint4 of LDS owned by the calling wave */
template <bool ADDR_LDS>
__device__ __forceinline__ void load_fft2_regs(Fft2Regs &R, const sea_fft_tables *t, int lane, uint4 *addrLds)
{
    const int j = lane & 31;
    const unsigned half = (unsigned)(lane >> 5) * 1024u; /* second transform: the next 256 words */
    const unsigned both = half | (half << 16);
#pragma unroll
    for (int s = 1; s < SEA_FFT_LSTAGES; ++s) { /* level 0 (n2 = 8) runs on registers: rfft256_head8 */
        R.kind[s] = t->fft2Item[s][j] >> 16;
        if (ADDR_LDS)
            addrLds[s * 64 + lane] = make_uint4(t->fft2Addr[s][0][j] + both, t->fft2Addr[s][1][j] + both,
                                                t->fft2Addr[s][2][j] + both, t->fft2Addr[s][3][j] + both);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ADDR_LDS) R.addr[s][k] = t->fft2Addr[s][k][j] + both;
            R.tw[s][k] = t->fft2Tw[s][k][j];
        }
    }
    R.addrLds = addrLds + lane;
    R.headA[0] = t->fft2Head[0][lane];
    R.headA[1] = t->fft2Head[1][lane];
    R.psdA[0] = t->fft2Psd[0][lane];
    R.psdA[1] = t->fft2Psd[1][lane];
    R.nyq = t->fft2Nyq;
    R.head8Flags = (SEA_FFT_HEAD16 && ADDR_LDS) ? t->fft16Flags[lane] : t->fft8Flags[lane];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        R.head8[q] = t->fft8Addr[q][lane] + both;
        R.head16[q] = t->fft16Addr[q][lane] + both;
        R.tw16[q] = t->fft2Tw[1][q][0]; /* level n2 = 16 has one twiddled item per block, j = 1: slot 0 is one of them */
    }
}

/* One split-radix level of the dual transform in three pieces, so that the caller can put the NEXT level's operand
 * reads right behind this level's result stores and keep the stored registers alive across them: a wave's LDS
 * operations execute in order, so the reads see the stores without any wait -- but when a read's destination register is
 * also the data register of a store still in flight the compiler must wait for that store first (it did: one
 * s_waitcnt per read, i.e. a second LDS latency per level).  fft2_keep() after the reads makes the allocator give the
 * reads registers of their own. */
struct Fft2Ops {
    float x[8];
    unsigned a01, a23, b01, b23;
};

template <int S, bool ADDR_LDS>
__device__ __forceinline__ void fft2_load(const float *work, const Fft2Regs &R, Fft2Ops &o)
{
    if (ADDR_LDS) {
        const uint4 ad = R.addrLds[S * 64];
        o.a01 = ad.x, o.a23 = ad.y, o.b01 = ad.z, o.b23 = ad.w;
    } else {
        o.a01 = R.addr[S][0], o.a23 = R.addr[S][1], o.b01 = R.addr[S][2], o.b23 = R.addr[S][3];
    }
    o.x[0] = fft_at(work, o.a01 & 0xffffu), o.x[1] = fft_at(work, o.a01 >> 16);
    o.x[2] = fft_at(work, o.a23 & 0xffffu), o.x[3] = fft_at(work, o.a23 >> 16);
    o.x[4] = fft_at(work, o.b01 & 0xffffu), o.x[5] = fft_at(work, o.b01 >> 16);
    o.x[6] = fft_at(work, o.b23 & 0xffffu), o.x[7] = fft_at(work, o.b23 >> 16);
}

/* keeps eight values in registers up to this point and orders the LDS accesses around it (no instruction) */
__device__ __forceinline__ void fft2_keep(const float (&v)[8])
{
    asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]) : "memory");
}

/* Both butterfly kinds are evaluated by every lane and the lane's own kind selects the results: a wave executes both
 * sides of a divergent branch anyway, and without the branch the two independent dependency chains interleave and the
 * compiler sees that the operand loads have been consumed (with the branch it had to protect the next level's loads
 * against the -- impossible -- path on which nobody consumed them: one s_waitcnt per load). */
template <int S>
__device__ __forceinline__ void fft2_butterfly(const Fft2Regs &R, const Fft2Ops &in, float (&o)[8])
{
    const bool tw = R.kind[S] == SEA_BF_TWIDDLE;
    const float x1 = in.x[0], x2 = in.x[1], x3 = in.x[2], x4 = in.x[3], x5 = in.x[4], x6 = in.x[5], x7 = in.x[6], x8 = in.x[7];
    float o1, o2, o3, o4, o5, o6, o7, o8;
    { /* SEA_BF_TWIDDLE, rfft.c:145-174 */
        const float cc1 = R.tw[S][0], ss1 = R.tw[S][1], cc3 = R.tw[S][2], ss3 = R.tw[S][3];
#if SEA_FFT_PACKED
        /* the same 24 operations on register pairs (v_pk_mul_f32 / v_pk_add_f32: each half rounded like the
         * scalar instruction; a - b == a + (-b) and the operand swaps / sign flips are instruction modifiers):
         *   (t1,t2) = (x3,x7)*(cc1,cc1) + (x7,x3)*(ss1,-ss1)      (t3,t4) likewise from (x4,x8), cc3, ss3 */
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f a37 = {x3, x7}, a48 = {x4, x8}, w1 = {cc1, ss1}, w3 = {cc3, ss3};
        /* the (c,c) and (s,-s) operands are selected out of the (c,s) pair by the instruction's op_sel / neg_hi
         * fields; written as asm because the compiler otherwise materialises them (8 more registers per level) */
        v2f p1, q1, p3, q3;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p1) : "v"(a37), "v"(w1));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(q1) : "v"(a37), "v"(w1));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p3) : "v"(a48), "v"(w3));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(q3) : "v"(a48), "v"(w3));
        const v2f T12 = p1 + q1, T34 = p3 + q3;
        const v2f T56 = T12 + T34, D34 = T12 - T34; /* (t5,t6), new (t3,t4) */
        const v2f o16 = v2f{x1, x1} + v2f{T56.x, -T56.x}; /* x1 + t5, x1 - t5 */
        const v2f o25 = v2f{x5, x5} + v2f{D34.y, -D34.y}; /* x5 + t4, x5 - t4 */
        const v2f o83 = v2f{T56.y, T56.y} + v2f{x6, -x6}; /* x6 + t6, t6 - x6 */
        const v2f o47 = v2f{x2, -x2} - v2f{D34.x, D34.x}; /* x2 - t3, -x2 - t3 */
        o1 = o16.x, o6 = o16.y, o2 = o25.x, o5 = o25.y, o8 = o83.x, o3 = o83.y, o4 = o47.x, o7 = o47.y;
#else
        float t1 = x3 * cc1 + x7 * ss1;
        float t2 = x7 * cc1 - x3 * ss1;
        float t3 = x4 * cc3 + x8 * ss3;
        float t4 = x8 * cc3 - x4 * ss3;
        const float t5 = t1 + t3, t6 = t2 + t4;
        t3 = t1 - t3;
        t4 = t2 - t4;
        o3 = t6 - x6;
        o8 = x6 + t6;
        o7 = -x2 - t3;
        o4 = x2 - t3;
        o6 = x1 - t5;
        o1 = x1 + t5;
        o5 = x5 - t4;
        o2 = x5 + t4;
#endif
    }
    float p1, p3, p4, p5, p6, p7, p8;
    { /* SEA_BF_PAIR: plain butterfly on the a-quadruple (rfft.c:110-113), pi/4 butterfly on
         the b-quadruple (rfft.c:120-125; the exact multiply form proven in fft_level) */
        const float t1 = x4 + x3;
        p4 = x4 - x3;
        p3 = x1 - t1;
        p1 = x1 + t1;
        const float u1 = (float)((double)(x7 + x8) * 0.70710678118654752440);
        const float u2 = (float)((double)(x7 - x8) * 0.70710678118654752440);
        p8 = x6 - u1;
        p7 = -x6 - u1;
        p6 = x5 - u2;
        p5 = x5 + u2;
    }
    o[0] = tw ? o1 : p1, o[1] = tw ? o2 : x2, o[2] = tw ? o3 : p3, o[3] = tw ? o4 : p4;
    o[4] = tw ? o5 : p5, o[5] = tw ? o6 : p6, o[6] = tw ? o7 : p7, o[7] = tw ? o8 : p8;
}

/* The throughput-bound users of the transform (rfft256, CompCeps, IRM: many waves per SIMD) keep the butterfly kinds as
 * a divergent branch: both sides still issue, but without the eight selects and the extra live registers of the
 * branch-free form (CompCeps 0.87 -> 1.09 ms with the branch-free form). */
__device__ __forceinline__ void fft2_bf_kind(const unsigned kind, const float (&tw)[4], const float (&x)[8], float (&o)[8])
{
    const float x1 = x[0], x2 = x[1], x3 = x[2], x4 = x[3], x5 = x[4], x6 = x[5], x7 = x[6], x8 = x[7];
    float o1, o2, o3, o4, o5, o6, o7, o8;
    if (kind == SEA_BF_TWIDDLE) { /* rfft.c:145-174 */
        const float cc1 = tw[0], ss1 = tw[1], cc3 = tw[2], ss3 = tw[3];
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f a37 = {x3, x7}, a48 = {x4, x8}, w1 = {cc1, ss1}, w3 = {cc3, ss3};
        v2f p1, q1, p3, q3;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p1) : "v"(a37), "v"(w1));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(q1) : "v"(a37), "v"(w1));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p3) : "v"(a48), "v"(w3));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(q3) : "v"(a48), "v"(w3));
        const v2f T12 = p1 + q1, T34 = p3 + q3;
        const v2f T56 = T12 + T34, D34 = T12 - T34;
        const v2f o16 = v2f{x1, x1} + v2f{T56.x, -T56.x};
        const v2f o25 = v2f{x5, x5} + v2f{D34.y, -D34.y};
        const v2f o83 = v2f{T56.y, T56.y} + v2f{x6, -x6};
        const v2f o47 = v2f{x2, -x2} - v2f{D34.x, D34.x};
        o1 = o16.x, o6 = o16.y, o2 = o25.x, o5 = o25.y, o8 = o83.x, o3 = o83.y, o4 = o47.x, o7 = o47.y;
    } else { /* SEA_BF_PAIR (rfft.c:110-113, :120-125) */
        const float t1 = x4 + x3;
        o4 = x4 - x3;
        o3 = x1 - t1;
        o1 = x1 + t1;
        o2 = x2;
        const float u1 = (float)((double)(x7 + x8) * 0.70710678118654752440);
        const float u2 = (float)((double)(x7 - x8) * 0.70710678118654752440);
        o8 = x6 - u1;
        o7 = -x6 - u1;
        o6 = x5 - u2;
        o5 = x5 + u2;
    }
    o[0] = o1, o[1] = o2, o[2] = o3, o[3] = o4, o[4] = o5, o[5] = o6, o[6] = o7, o[7] = o8;
}
template <int S>
__device__ __forceinline__ void fft2_butterfly_branchy(const Fft2Regs &R, const Fft2Ops &in, float (&o)[8])
{
    fft2_bf_kind(R.kind[S], R.tw[S], in.x, o);
}

template <int S>
__device__ __forceinline__ void fft2_store(float *work, const Fft2Regs &R, const Fft2Ops &in, const float (&o)[8])
{
    if (R.kind[S] != SEA_BF_NONE) {
        fft_at(work, in.a01 & 0xffffu) = o[0];
        fft_at(work, in.a01 >> 16) = o[1];
        fft_at(work, in.a23 & 0xffffu) = o[2];
        fft_at(work, in.a23 >> 16) = o[3];
        fft_at(work, in.b01 & 0xffffu) = o[4];
        fft_at(work, in.b01 >> 16) = o[5];
        fft_at(work, in.b23 & 0xffffu) = o[6];
        fft_at(work, in.b23 >> 16) = o[7];
    }
}

/* one complete level (load, butterfly, store), throughput form */
template <int S, bool ADDR_LDS>
__device__ __forceinline__ void fft2_level(float *work, const Fft2Regs &R)
{
    Fft2Ops in;
    float o[8];
    fft2_load<S, ADDR_LDS>(work, R, in);
    fft2_butterfly_branchy<S>(R, in, o);
    fft2_store<S>(work, R, in, o);
}

/* the register-resident start of rfft256 (bit reversal, length-2 and n2=4 butterflies) for one
 * frame; stores the lane's four values at the (swizzled) places of elements 4r..4r+3, r = bitrev6(lane) */
__device__ __forceinline__ void rfft256_head(float e0, float e1, float e2, float e3, float *work,
                                             unsigned flags, const unsigned (&head)[2])
{
    float g0 = e0, g1 = e2, g2 = e1, g3 = e3;
    {
        const float s01 = g0 + g1, d01 = g0 - g1, s23 = g2 + g3, d23 = g2 - g3;
        const bool f0 = (flags & 1u) != 0, f1 = (flags & 2u) != 0;
        g0 = f0 ? s01 : g0;
        g1 = f0 ? d01 : g1;
        g2 = f1 ? s23 : g2;
        g3 = f1 ? d23 : g3;
    }
    {
        const float t1 = g3 + g2;
        const float n3 = g3 - g2, n2 = g0 - t1, n0 = g0 + t1;
        const bool f = (flags & 4u) != 0;
        g3 = f ? n3 : g3;
        g2 = f ? n2 : g2;
        g0 = f ? n0 : g0;
    }
    fft_at(work, head[0] & 0xffffu) = g0;
    fft_at(work, head[0] >> 16) = g1;
    fft_at(work, head[1] & 0xffffu) = g2;
    fft_at(work, head[1] >> 16) = g3;
}

/* two transforms at once: eA / eB hold the lane's four (windowed) elements of frame A / B.
 * The transform is offered in two halves so that a pipelined kernel can run them in different
 * waves (one frame apart): _lo = register-resident start (up to n2 = 8) + levels n2 = 16, 32; _hi = levels
 * n2 = 64, 128, 256.  Both end with wave_sync(). */
/* The register-resident start of BOTH transforms: lane l holds in e[0..7] the (windowed) input elements
 * n0 + 32 * bitrev3(j), n0 = l & 31, of its transform (l >> 5), i.e. the bit-reversed positions 8k..8k+7,
 * k = bitrev5(n0).  The length-2 butterflies (rfft.c:82-96), the n2 = 4 level (:100-113, plain butterflies
 * only) and the n2 = 8 level (:100-125, plain + pi/4 butterflies, no twiddles yet) touch nothing outside
 * such a block, so all three run on registers, gated per block by the reference's is/id schedule
 * (fft8Flags); the eight results go to their swizzled places in this lane's work area. */
__device__ __forceinline__ void rfft256_head8_regs(float (&e)[8], const unsigned fl)
{
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = e[2 * p], b = e[2 * p + 1], sum = a + b, dif = a - b;
        const bool f = (fl & (1u << p)) != 0;
        e[2 * p] = f ? sum : a;
        e[2 * p + 1] = f ? dif : b;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float g0 = e[4 * h], g2 = e[4 * h + 2], g3 = e[4 * h + 3];
        const float t1 = g3 + g2;
        const float n3 = g3 - g2, n2 = g0 - t1, n0 = g0 + t1;
        const bool f = (fl & (16u << h)) != 0;
        e[4 * h + 3] = f ? n3 : g3;
        e[4 * h + 2] = f ? n2 : g2;
        e[4 * h] = f ? n0 : g0;
    }
    {
        /* a-quadruple = positions 8k + {0,2,4,6} (n4 = 2), b-quadruple = 8k + {1,3,5,7} (n8 = 1) */
        const float x1 = e[0], x2 = e[2], x3 = e[4], x4 = e[6], x5 = e[1], x6 = e[3], x7 = e[5], x8 = e[7];
        const float t1 = x4 + x3;
        const float o4 = x4 - x3, o3 = x1 - t1, o1 = x1 + t1;
        const float u1 = (float)((double)(x7 + x8) * 0.70710678118654752440);
        const float u2 = (float)((double)(x7 - x8) * 0.70710678118654752440);
        const float o8 = x6 - u1, o7 = -x6 - u1, o6 = x5 - u2, o5 = x5 + u2;
        const bool f = (fl & 64u) != 0;
        e[0] = f ? o1 : x1;
        e[2] = x2;
        e[4] = f ? o3 : x3;
        e[6] = f ? o4 : x4;
        e[1] = f ? o5 : x5;
        e[3] = f ? o6 : x6;
        e[5] = f ? o7 : x7;
        e[7] = f ? o8 : x8;
    }
}
/* SEA_FFT_HEAD16: ... and the n2 = 16 level (rfft.c:100-174 at n2 = 16; sea_tables.h fft16*).  Lane and lane ^ 16 hold the two halves
 * of a 16-block; the lower one gives its four odd positions and takes the upper one's four even ones (v_permlane16_swap), then runs the
 * block's PAIR item on the eight even positions while the upper one runs the twiddled item (j = 1) on the eight odd ones -- the
 * butterflies of fft2_bf_kind on registers.  A block outside the level's schedule passes through.  One LDS round trip (eight stores,
 * eight loads per lane) less than head8 + fft2_level<1>. */
__device__ __forceinline__ void rfft256_head16(float (&e)[8], float *work, const Fft2Regs &R)
{
    rfft256_head8_regs(e, R.head8Flags);
    const bool up = (threadIdx.x & 16u) != 0; /* the half at positions 16b + 8 .. 16b + 15 */
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float give = up ? e[2 * i] : e[2 * i + 1];
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(give), __float_as_uint(give), false, false);
        r[i] = __uint_as_float(up ? sw[0] : sw[1]); /* odd rows find the even row's value in [0], even rows the odd row's in [1] */
    }
    /* item order: a, a + n4, a + 2 n4, a + 3 n4, b, ...: lower half (PAIR) 0 4 8 12 | 2 6 10 14, upper half (j = 1) 1 5 9 13 | 3 7 11 15 */
    const float x[8] = {up ? r[0] : e[0], up ? r[2] : e[4], up ? e[1] : r[0], up ? e[5] : r[2],
                        up ? r[1] : e[2], up ? r[3] : e[6], up ? e[3] : r[1], up ? e[7] : r[3]};
    float o[8];
    fft2_bf_kind(up ? (unsigned)SEA_BF_TWIDDLE : (unsigned)SEA_BF_PAIR, R.tw16, x, o);
    const bool on = (R.head8Flags & 128u) != 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        fft_at(work, R.head16[q] & 0xffffu) = on ? o[2 * q] : x[2 * q];
        fft_at(work, R.head16[q] >> 16) = on ? o[2 * q + 1] : x[2 * q + 1];
    }
}
__device__ __forceinline__ void rfft256_head8(float (&e)[8], float *work, const Fft2Regs &R)
{
    rfft256_head8_regs(e, R.head8Flags);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        fft_at(work, R.head8[q] & 0xffffu) = e[2 * q];
        fft_at(work, R.head8[q] >> 16) = e[2 * q + 1];
    }
}

/* levels S0 .. S1 chained: level k+1's operand reads go out right behind level k's stores (see Fft2Ops) */
template <int S0, int S1, bool ADDR_LDS>
__device__ __forceinline__ void fft2_levels(float *work, const Fft2Regs &R, float (&prev)[8])
{
    Fft2Ops in;
    fft2_load<S0, ADDR_LDS>(work, R, in);
    fft2_keep(prev);
    float o[8];
    fft2_butterfly<S0>(R, in, o);
    fft2_store<S0>(work, R, in, o);
    wave_sync();
    if constexpr (S0 < S1) fft2_levels<S0 + 1, S1, ADDR_LDS>(work, R, o);
    else fft2_keep(o);
}
/* the same chain with level S1's results left in `last` instead of stored (rfft256_dual_keep_last) */
template <int S0, int S1, bool ADDR_LDS>
__device__ __forceinline__ void fft2_levels_keep_last(float *work, const Fft2Regs &R, float (&prev)[8], float (&last)[8])
{
    Fft2Ops in;
    fft2_load<S0, ADDR_LDS>(work, R, in);
    fft2_keep(prev);
    if constexpr (S0 < S1) {
        float o[8];
        fft2_butterfly<S0>(R, in, o);
        fft2_store<S0>(work, R, in, o);
        wave_sync();
        fft2_levels_keep_last<S0 + 1, S1, ADDR_LDS>(work, R, o, last);
    } else {
        fft2_butterfly<S0>(R, in, last);
    }
}

template <bool ADDR_LDS>
__device__ __forceinline__ void rfft256_dual_lo(float (&e)[8], float *work, const Fft2Regs &R)
{
    if (SEA_FFT_HEAD16 && ADDR_LDS) {
        rfft256_head16(e, work, R);
        wave_sync();
    } else {
        rfft256_head8(e, work, R);
        wave_sync();
        fft2_level<1, ADDR_LDS>(work, R);
        wave_sync();
    }
    fft2_level<2, ADDR_LDS>(work, R);
    wave_sync();
}

template <bool ADDR_LDS>
__device__ __forceinline__ void rfft256_dual_hi(float *work, const Fft2Regs &R)
{
    fft2_level<3, ADDR_LDS>(work, R);
    wave_sync();
    fft2_level<4, ADDR_LDS>(work, R);
    wave_sync();
    fft2_level<5, ADDR_LDS>(work, R);
    wave_sync();
}

/* LAT = true: the latency form (levels chained, branch-free butterflies) for a wave whose run time is its own chain of
 * dependent instructions -- the transform wave of the pipelined NoiseSup kernels; LAT = false: the throughput form */
template <bool ADDR_LDS, bool LAT = false>
__device__ __forceinline__ void rfft256_dual(float (&e)[8], float *work, const Fft2Regs &R)
{
    if (LAT) {
        if (SEA_FFT_HEAD16 && ADDR_LDS) {
            rfft256_head16(e, work, R);
            wave_sync();
            fft2_levels<2, 5, ADDR_LDS>(work, R, e);
        } else {
            rfft256_head8(e, work, R);
            wave_sync();
            fft2_levels<1, 5, ADDR_LDS>(work, R, e);
        }
    } else {
        rfft256_dual_lo<ADDR_LDS>(e, work, R);
        rfft256_dual_hi<ADDR_LDS>(work, R);
    }
}

/* second half of the dual transform with the last level's results left in registers (see rfft256_dual_keep_last) */
template <bool ADDR_LDS>
__device__ __forceinline__ void rfft256_dual_hi_keep_last(float *work, const Fft2Regs &R, float (&o)[8])
{
    fft2_level<3, ADDR_LDS>(work, R);
    wave_sync();
    fft2_level<4, ADDR_LDS>(work, R);
    wave_sync();
    Fft2Ops in;
    fft2_load<5, ADDR_LDS>(work, R, in);
    fft2_butterfly_branchy<5>(R, in, o);
}

/* The throughput form with the LAST level's results left in registers (no store): the lane's item of level n2 = 256 is, for
 * slots 0..30 (SEA_BF_TWIDDLE, j = slot + 1), o[0..7] = x[j], x[64+j], x[128+j], x[192+j], x[64-j], x[128-j], x[192-j], x[256-j]
 * -- four complete bins: (Re, Im) of j = (o0, o7), of 64+j = (o1, o6), of 64-j = (o4, o3), of 128-j = (o5, o2) -- and for slot 31
 * (SEA_BF_PAIR) x[0], x[64], x[128], x[192], x[32], x[96], x[160], x[224]: bins 0 = o0, 64 = (o1, o3), 128 = o2, 32 = (o4, o7),
 * 96 = (o5, o6) (sea_tables.c build_fft, half-wave items; rfft.c:100-174).  A caller that only needs |X|^2 saves the level's
 * eight stores, its own reads of the spectrum and one LDS round trip. */
template <bool ADDR_LDS>
__device__ __forceinline__ void rfft256_dual_keep_last(float (&e)[8], float *work, const Fft2Regs &R, float (&o)[8])
{
    rfft256_dual_lo<ADDR_LDS>(e, work, R);
    fft2_level<3, ADDR_LDS>(work, R);
    wave_sync();
    fft2_level<4, ADDR_LDS>(work, R);
    wave_sync();
    Fft2Ops in;
    fft2_load<5, ADDR_LDS>(work, R, in);
    fft2_butterfly_branchy<5>(R, in, o);
}

/* float -> int16 exactly as the reference's (short) cast behaves on x86-64: truncate toward zero
 * to int32 and keep the low 16 bits; out-of-int32-range gives 0 (etsi/cpp/ParmInterface.c:266). */
__device__ __forceinline__ int cast_i16(float v)
{
    const bool ok = (v > -2147483648.0f) && (v < 2147483648.0f);
    const int i = ok ? (int)v : 0;
    return i & 0xFFFF;
}

/* sum init + p[0] + p[1] + ... + p[n-1] in exactly that order (all lanes compute the same value).
 * p must be 16-byte aligned and padded to a multiple of 4 floats. */
template <int N>
__device__ __forceinline__ float serial_sum(const float *p, float init)
{
    float acc = init;
#pragma unroll 4 /* bounded: a full unroll hoists all N loads and costs N VGPRs */
    for (int i = 0; i < N / 4; ++i) {
        const float4 v = *reinterpret_cast<const float4 *>(p + 4 * i);
        acc += v.x;
        acc += v.y;
        acc += v.z;
        acc += v.w;
    }
#pragma unroll
    for (int i = N / 4 * 4; i < N; ++i) acc += p[i];
    return acc;
}

} // namespace sea
