/*
 * resynth_kernel.hip -- Hu-Wang 64-channel gammatone analysis/synthesis resynthesis, gfx950.
 *
 * One workgroup (three 64-lane wavefronts) owns one utterance; LANE = CHANNEL (64 channels = one
 * wave), so the 4th-order complex one-pole cascade of every channel advances one sample per step
 * in lock-step.  A wave that is alone on its SIMD issues one vector instruction every 4 cycles, so
 * the ~49 operations of one cascade step bound a one-wave recurrence at ~200 cycles per sample.  The
 * cascade is therefore cut in two feed-forward halves that run as a software pipeline, 16-sample
 * tiles through LDS, one s_barrier per tile:
 *
 *   wave R1  cascade stages 0-1 (state p0,q0,p1,q1)          -> (p1,q1) per sample
 *   wave R2  cascade stages 2-3 (state p2,q2,p3,q3; it re-derives x1,y1 from the previous
 *            (p1,q1), which it already holds)                -> filter output per sample
 *   wave H   everything per-sample but not recursive: divisions by the middle-ear gain, the
 *            overlap-add weights, channel sums, casts, HBM stores
 *
 *   resynth_fwd_kernel : analysis pass; H writes g1[n][c]/midEar[c] to HBM as rows of 64 floats
 *                        (256 B per step, fully coalesced).  288 GB of HBM is what makes keeping the
 *                        whole [L][64] intermediate of a 1024-utterance batch (~17 GB) resident
 *                        feasible.
 *   resynth_bwd_kernel : R1 reads the rows in reverse time order (8 in flight per lane); H divides
 *                        again, evaluates the mask-weighted raised-cosine overlap-add weight of
 *                        that sample on the fly (at most two overlapping frames per sample),
 *                        multiplies, and sums the 64 channels IN CHANNEL ORDER through a padded LDS
 *                        transpose (lane = sample), then truncates to int16.  No second
 *                        intermediate is written.
 *
 * Reference reproduced: resyth_64sub_ori/cpp/extractwav.cpp:55-121 (resynth body; hairCell is dead
 * code there), :167-211 (gammaToneFilter); resyth_64sub_IBM/cpp/extractwav.cpp:97-99
 * (binary mask).  Float arithmetic order is the reference's (-ffp-contract=off).
 */
#include "sea_device.h"
#include "sea_kernels.h"

namespace sea {

namespace {


/* timing-only diagnostic (-DSEA_RS_TIMING): per role, shader-clock cycles spent working and waiting
 * at the tile barrier, for workgroup 0 -> g_rs_timing[kernel*6 + role*2 + {0,1}] */
#ifdef SEA_RS_TIMING
__device__ unsigned g_rs_wg[4096 * 4]; /* fused kernel, per workgroup: start and end on the constant 100 MHz counter (low 32 bits), HW_ID, XCC_ID */
__device__ unsigned long long g_rs_timing[32]; /* fwd R1,R2,R3 = 0..5; bwd R1,R2,W,SUM = 6..13; subband R1,R2,K,HC,W = 16..25 */
struct RoleTimer {
    unsigned long long work = 0, wait = 0, t0 = 0, t1 = 0;
    __device__ __forceinline__ void begin() { t0 = clock64(); }
    __device__ __forceinline__ void mid() { t1 = clock64(); work += t1 - t0; }
    __device__ __forceinline__ void end() { wait += clock64() - t1; }
    __device__ __forceinline__ void flush(int slot)
    {
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
            g_rs_timing[slot] = work;
            g_rs_timing[slot + 1] = wait;
        }
    }
};
#define RS_T_DECL RoleTimer rt_
#define RS_T_BEGIN rt_.begin()
#define RS_T_MID rt_.mid()
#define RS_T_END rt_.end()
#define RS_T_FLUSH(slot) rt_.flush(slot)
#else
#define RS_T_DECL
#define RS_T_BEGIN
#define RS_T_MID
#define RS_T_END
#define RS_T_FLUSH(slot)
#endif

/* One sample of gammaToneFilter (extractwav.cpp:188-210) is
 *     out = p3*gain (taken BEFORE the update);  x_k = f1 p_k - f2 q_k,  y_k = f2 p_k + f1 q_k;
 *     p0 = in*f1 + x0;  p1 = p0 + x1;  p2 = p1 + x1 + x2;  p3 = p2 + x1 + 2 x2 + x3   (q alike with y).
 * It is evaluated in two halves (stages 0-1, stages 2-3) so that two waves can pipeline it. */

/* Both halves work on PACKED pairs: the state of cascade stage k is the pair (p_k, q_k) and
 * every operation of the reference acts on p and q alike, so each pair lives in an even-aligned
 * VGPR pair and is advanced by v_pk_mul_f32 / v_pk_add_f32 (two IEEE float operations per
 * instruction, individually rounded exactly like v_mul_f32 / v_add_f32; -ffp-contract=off keeps
 * them unfused).  The rotation  x = f1 p - f2 q,  y = f2 p + f1 q  becomes
 *     (f1,f1)*(p,q) + (-f2,f2)*(q,p)
 * -- negating f2 is exact and float addition commutes, so x and y are bit-identical to the
 * reference's expressions; the (q,p) swap is an operand select of the packed instruction.  This
 * halves the vector instructions of the recurrence (22 instead of 43 per sample and channel). */
typedef float v2f __attribute__((ext_vector_type(2)));

struct GtCoef {
    v2f f11, nf2, f12; /* (f1,f1), (-f2,f2), (f1,f2) */
};
__device__ __forceinline__ GtCoef gt_coef(float f1, float f2)
{
    GtCoef c;
    c.f11 = v2f{f1, f1};
    c.nf2 = v2f{-f2, f2};
    c.f12 = v2f{f1, f2};
    return c;
}
__device__ __forceinline__ v2f gt_rot(const GtCoef &c, v2f s) { return c.f11 * s + c.nf2 * s.yx; }

/* Stages 0-1: consumes the input sample, returns the NEW (p1,q1). */
struct GtLo {
    v2f s0, s1;
};
/* in2 = (in, in): callers pass a .xx / .yy swizzle of the register PAIR the sample already sits in,
 * which the packed multiply takes as an operand select -- a lone float would first have to be moved
 * into the low half of an even-aligned pair. */
__device__ __forceinline__ v2f gt_step_lo(GtLo &s, v2f in2, const GtCoef &c)
{
    const v2f r0 = gt_rot(c, s.s0), r1 = gt_rot(c, s.s1);
    s.s0 = in2 * c.f12 + r0; /* p0 = in*f1 + x0, q0 = in*f2 + y0 */
    s.s1 = s.s0 + r1;
    return s.s1;
}
/* Stages 2-3: needs the new (p1,q1) and x1,y1 -- the rotation of the OLD (p1,q1), which this wave
 * kept from the previous sample, so the same products and sum reproduce them exactly. */
struct GtHi {
    v2f s1old, s2, s3;
};
__device__ __forceinline__ float gt_step_hi(GtHi &s, v2f pq1, const GtCoef &c, float gain)
{
    const float out = s.s3.x * gain;
    const v2f r1 = gt_rot(c, s.s1old), r2 = gt_rot(c, s.s2), r3 = gt_rot(c, s.s3);
    s.s2 = pq1 + r1 + r2;
    s.s3 = s.s2 + r1 + 2.0f * r2 + r3;
    s.s1old = pq1;
    return out;
}


/* workgroup barrier for the role-specialised waves (same count, different program counters);
 * LDS-only fences: HBM loads and stores stay in flight across it */
__device__ __forceinline__ void tile_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

/* Issue priority by REMAINING tiles (longest-remaining-processing-time first, as the four-wave NoiseSup kernel has it,
 * ns_pipe_kernel.hip: SEA_PRIO_LRPT): every wave counts its tile barriers down and, every 64 tiles, sets s_setprio from
 * 32 * remaining / (the batch's longest utterance) dithered over eight consecutive evaluations into the four hardware
 * levels, so that the utterances sharing a CU converge on a common finishing time.  scale = 0: off. */
#ifndef SEA_RS_FEED_W3
#define SEA_RS_FEED_W3 1
#endif
#ifndef SEA_RS_LRPT
#define SEA_RS_LRPT 1
#endif
struct TilePrio {
    long long left; /* tile barriers this wave still has to pass */
    float scale;    /* 32 / (tile barriers of the batch's longest utterance); 0: off */
    __device__ __forceinline__ void tick()
    {
        --left;
        if (SEA_RS_LRPT && scale > 0.0f && (left & 63) == 0) {
            const int lv = __builtin_amdgcn_readfirstlane((int)((float)left * scale)) + (int)((left >> 6) & 7);
            if (lv >= 24) __builtin_amdgcn_s_setprio(3);
            else if (lv >= 16) __builtin_amdgcn_s_setprio(2);
            else if (lv >= 8) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
    }
};
__device__ __forceinline__ void tile_sync(TilePrio &p)
{
    tile_sync();
    p.tick();
}

constexpr int kTile = 16;        /* time steps per hand-over between the pipelined waves */
constexpr int kMacro = 2;        /* tiles per HBM request group of the synthesis pass */
constexpr int kTileStride = 68;  /* floats per step in the padded tile: 64 channels + 4 (rows stay 16-byte aligned
                                  * and 16 lanes reading 16 different rows as float4 hit 64 different banks) */

/* ---- division by the per-channel middle-ear gain ------------------------------------------------
 * Both passes divide every sample of every channel by midEar[c] (extractwav.cpp:86-90).  An IEEE
 * float division expands to ~10 instructions; with the divisor a per-lane constant d and
 * y = RN(1/d) computed once,
 *     q = RN(a y);  r = RN(a - q d) (exact, one fma);  q' = RN(q + r y)
 * returns RN(a/d) (Markstein's correction step) as long as nothing underflows.  This is not taken
 * on faith: sea_selftest_div() runs all 2^32 float patterns of a through it for each of the 64
 * divisors and compares with the hardware-correct a/d bit for bit; the kernels use it only inside
 * the verified domain 2^-100 <= |a| <= 2^100 (checked per tile with one min3 and one max3 per two
 * samples) and redo a tile with true divisions otherwise (zeros, denormal filter tails). */
struct DivConst {
    float d, y;
};
__device__ __forceinline__ DivConst div_const(float d)
{
    DivConst c;
    c.d = d;
    c.y = 1.0f / d;
    return c;
}
__device__ __forceinline__ float div_fast(float a, const DivConst &c)
{
    const float q = a * c.y;
    const float r = __fmaf_rn(-q, c.d, a);
    return __fmaf_rn(r, c.y, q);
}
__device__ __forceinline__ v2f div_fast2(v2f a, const DivConst &c)
{
    const v2f y = {c.y, c.y}, nd = {-c.d, -c.d};
    const v2f q = a * y;
    const v2f r = __builtin_elementwise_fma(q, nd, a);
    return __builtin_elementwise_fma(r, y, q);
}
constexpr float kDivLo = 0x1p-100f, kDivHi = 0x1p100f;
__device__ __forceinline__ bool div_in_domain(float a)
{
    const float aa = fabsf(a);
    return aa >= kDivLo && aa <= kDivHi;
}

/* 16 steps of one channel divided by its constant: packed fast path, with the domain check folded
 * into one v_max3 / v_min3 per pair (inline asm: the |x| operand modifiers are free, and the
 * library fmaxf would add a canonicalisation per operand); the whole wave redoes the tile with true
 * divisions if any lane has a sample outside the verified domain (wave-uniform, rare). */
__device__ __forceinline__ void div_tile16(const float (&g)[kTile], float (&v)[kTile], const DivConst &c)
{
    float mx = 0.0f, mn = kDivHi;
#pragma unroll
    for (int k = 0; k < kTile; k += 2) {
        const v2f q = div_fast2(v2f{g[k], g[k + 1]}, c);
        v[k] = q.x;
        v[k + 1] = q.y;
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(mx) : "v"(mx), "v"(g[k]), "v"(g[k + 1]));
        asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(mn) : "v"(mn), "v"(g[k]), "v"(g[k + 1]));
    }
    const bool outside = !(mn >= kDivLo && mx <= kDivHi);
    if (__ballot(outside) != 0ull) {
#pragma unroll
        for (int k = 0; k < kTile; ++k) v[k] = g[k] / c.d;
    }
}

/* int16 input of one utterance for the R1 wave (lane = channel needs every sample broadcast): one
 * coalesced 128-byte request fetches a CHUNK of 64 samples = 4 tiles, raw, three chunks ahead of
 * use, so that no HBM latency is ever waited for inside the per-tile loop (the loads are
 * unconditional -- clamped address, value selected afterwards -- because a load under a branch
 * gets its s_waitcnt at the end of that branch). */
struct InFeed {
    int c0, c1, c2; /* raw samples of chunks q, q+1, q+2 (this lane's sample of each) */
    static __device__ __forceinline__ int fetch(const int16_t *in, long long L, long long chunk, int lane)
    {
        long long i = chunk * 64 + lane;
        i = (i < L) ? i : L - 1;
        return (int)in[i];
    }
    __device__ __forceinline__ void start(const int16_t *in, long long L, int lane)
    {
        c0 = c1 = c2 = 0;
        if (L > 0) {
            c0 = fetch(in, L, 0, lane);
            c1 = fetch(in, L, 1, lane);
            c2 = fetch(in, L, 2, lane);
        }
    }
    /* deposit the 16 samples of tile j in xs[0..15] (zeros beyond L); L > 0 here */
    __device__ __forceinline__ void tile(const int16_t *in, long long L, long long j, int lane, float *xs)
    {
        const int q = (int)(j & 3);
        if (q == 0 && j > 0) {
            c0 = c1;
            c1 = c2;
            c2 = fetch(in, L, (j >> 2) + 2, lane);
        }
        if ((lane >> 4) == q) {
            const long long n = j * kTile + (lane & 15);
            xs[lane & 15] = (n < L) ? (float)c0 : 0.0f;
        }
    }
};

/* Stages 2-3 cut once more for the analysis pass (its LDS has room for the extra hand-over):
 * stage 2 alone returns A = (s2' + r1) + 2 r2, the reference's partial sum of p3/q3 in its own order;
 * stage 3 finishes s3' = A + r3 and emits p3*gain. */
struct GtMid {
    v2f s1old, s2;
};
__device__ __forceinline__ v2f gt_step_mid(GtMid &s, v2f pq1, const GtCoef &c)
{
    const v2f r1 = gt_rot(c, s.s1old), r2 = gt_rot(c, s.s2);
    s.s2 = pq1 + r1 + r2;
    s.s1old = pq1;
    return s.s2 + r1 + 2.0f * r2;
}
__device__ __forceinline__ float gt_step_top(v2f &s3, v2f A, const GtCoef &c, float gain)
{
    const float out = s3.x * gain;
    s3 = A + gt_rot(c, s3);
    return out;
}

/* HBM intermediate between the analysis and the synthesis pass: per utterance ntile = ceil(L/16)
 * tiles of 16 time steps x 64 channels, tile = 4 KB = four 1 KB quarters, quarter k holding steps
 * 4k..4k+3 of every channel as one float4 per lane (lane = channel).  Both passes move a tile with
 * four wave-wide 16-byte-per-lane accesses, each of them one fully contiguous kilobyte.  Utterance
 * u starts at float (offsets[u] + 8u) * 64: offsets are multiples of 8 samples, so the extra 8
 * steps per utterance make room for rounding L up to whole tiles (sea_resynth_scratch_bytes). */
__device__ __forceinline__ float *inter_base(float *inter, long long off, int u)
{
    return inter + (off + 8LL * u) * 64;
}

struct __attribute__((aligned(16))) RsLds {
    v2f pq[2][kTile][64];  /* R1 -> R2 */
    float g[2][kTile][64];   /* R2 -> H  */
};

} // namespace

/* Analysis pass: three waves per utterance (R1 stages 0-1 | R2 stage 2 | R3 stage 3, division by the
 * middle-ear gain, HBM store), 16-step tiles, one barrier per tile. */
namespace {

struct __attribute__((aligned(16))) FwdLds {
    v2f pq[2][kTile][64]; /* R1 -> R2: new (p1,q1) */
    v2f pa[2][kTile][64]; /* R2 -> R3: partial sums A */
    float xs[2][kTile]; /* SEA_RS_FEED_W3: the input samples of tile j at [j & 1], deposited one tile ahead by the fourth wave */
};

/* roles 0..2 work; any further wave of the workgroup only keeps the barrier count (fused kernel) */
__device__ __forceinline__ void resynth_fwd_body(const ResynthArgs &a, FwdLds &S, int role, int lane, int u,
                                                 long long off, long long L, TilePrio &tp)
{
    v2f(*pq)[kTile][64] = S.pq;
    v2f(*pa)[kTile][64] = S.pa;
    const long long ntile = (L + kTile - 1) / kTile, niter = ntile + 2;
    const int nwaves = (int)(blockDim.x >> 6);
    RS_T_DECL;
    if (role > 2) {
        /* SEA_RS_FEED_W3 (round 4): the fused kernel's fourth wave, idle during this pass, converts and deposits the input samples of
         * tile j + 1 while the cascade works on tile j: the first cascade wave R1 -- the longest role of the pass -- no longer pays a
         * store / fence / load round trip per tile for them */
        const int16_t *in = a.in + off;
        InFeed feed;
        if (SEA_RS_FEED_W3 && role == 3) feed.start(in, L, lane);
        for (long long j = 0; j < niter; ++j) {
            if (SEA_RS_FEED_W3 && role == 3 && j + 1 < ntile) feed.tile(in, L, j + 1, lane, S.xs[(j + 1) & 1]);
            tile_sync(tp);
        }
    } else if (role == 0) {
        const int16_t *in = a.in + off;
        const GtCoef C = gt_coef(a.tables->f1[lane], a.tables->f2[lane]);
        GtLo s = {};
        InFeed feed;
        feed.start(in, L, lane); /* extractwav.cpp:55-58 */
        const bool ownFeed = !(SEA_RS_FEED_W3 && nwaves > 3); /* the three-wave kernel of the split form feeds itself */
        for (long long j = 0; j < niter; ++j) {
            RS_T_BEGIN;
            if (j < ntile) {
                float *xs = S.xs[j & 1];
                if (ownFeed || j == 0) feed.tile(in, L, j, lane, xs);
                wave_sync();
                v2f(*o)[64] = pq[j & 1];
#pragma unroll
                for (int t = 0; t < kTile; t += 2) {
                    const v2f x2 = *reinterpret_cast<const v2f *>(&xs[t]);
                    o[t][lane] = gt_step_lo(s, x2.xx, C);
                    o[t + 1][lane] = gt_step_lo(s, x2.yy, C);
                }
                wave_sync();
            }
            RS_T_MID;
            tile_sync(tp);
            RS_T_END;
        }
        RS_T_FLUSH(0);
    } else if (role == 1) {
        const GtCoef C = gt_coef(a.tables->f1[lane], a.tables->f2[lane]);
        GtMid s = {};
        for (long long j = 0; j < niter; ++j) {
            RS_T_BEGIN;
            const long long jt = j - 1;
            if (jt >= 0 && jt < ntile) {
                const v2f(*i)[64] = pq[jt & 1];
                v2f(*o)[64] = pa[jt & 1];
#pragma unroll
                for (int t = 0; t < kTile; ++t) o[t][lane] = gt_step_mid(s, i[t][lane], C);
            }
            RS_T_MID;
            tile_sync(tp);
            RS_T_END;
        }
        RS_T_FLUSH(2);
    } else {
        /* stage 3, then reverse[...] = gOut / midEar (extractwav.cpp:86-87), streamed to HBM tile by
         * tile in the layout of inter_base(): four fully contiguous 1 KB stores per tile.  The last
         * tile is written whole; its steps past L carry the filter's response to the zero padding
         * and are never consumed (the synthesis pass substitutes zeros there). */
        const GtCoef C = gt_coef(a.tables->f1[lane], a.tables->f2[lane]);
        const float gain = a.tables->gain[lane];
        float *dst = inter_base(a.inter, off, u) + lane * 4;
        const DivConst ear = div_const(a.tables->midEar[lane]);
        v2f s3 = {0.0f, 0.0f};
        for (long long j = 0; j < niter; ++j) {
            RS_T_BEGIN;
            const long long jt = j - 2;
            if (jt >= 0 && jt < ntile) {
                const v2f(*i)[64] = pa[jt & 1];
                float *row = dst + jt * (kTile * 64);
                float gv[kTile], v[kTile];
#pragma unroll
                for (int t = 0; t < kTile; ++t) gv[t] = gt_step_top(s3, i[t][lane], C, gain);
                div_tile16(gv, v, ear);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<float4 *>(row + k * 256) = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
            }
            RS_T_MID;
            tile_sync(tp);
            RS_T_END;
        }
        RS_T_FLUSH(4);
    }
}

} // namespace

__global__ __launch_bounds__(192, 3) void resynth_fwd_kernel(ResynthArgs a)
{
    __shared__ FwdLds S;
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    TilePrio tp = {0, 0.0f};
    resynth_fwd_body(a, S, role, lane, u, a.offsets[u], a.lengths[u], tp);
}

/* gammaToneFilter() for one channel of the bank (HuWang.h:49): a serial recurrence, one lane. */
__global__ __launch_bounds__(64) void gammatone_kernel(const float *in, float *out, int chan, long long L,
                                                       const sea_gt_tables *t)
{
    if (threadIdx.x != 0) return;
    const GtCoef C = gt_coef(t->f1[chan], t->f2[chan]);
    const float gain = t->gain[chan];
    GtLo lo = {};
    GtHi hi = {};
    for (long long n = 0; n < L; ++n) {
        const v2f pq1 = gt_step_lo(lo, v2f{in[n], in[n]}, C);
        out[n] = gt_step_hi(hi, pq1, C, gain);
    }
}

/* Synthesis pass: FOUR waves per utterance.
 *   R1  reads the stored tiles back to front (HBM), cascade stages 0-1
 *   R2  cascade stages 2-3 -> filter output g into a padded tile
 *   W   per step and channel: the mask-weighted raised-cosine overlap-add weight of that sample (at most
 *       two overlapping frames) times g / midEar, written back IN PLACE
 *   SUM two jobs, because the 64-deep channel sum alone leaves this wave idle 60 % of a tile period while
 *       W was the longest role: (1) lane = channel, g / midEar of the tile R2 has just finished, in place
 *       (taken over from W: role timers W 2256 -> ~1700 clk per tile); (2) lane = step, the 64 channel
 *       terms of the tile W finished added in channel order, (short) cast, store
 * The tile passes through four owners (R2, division, W, sum), hence four buffers. */
namespace {

struct __attribute__((aligned(16))) BwdLds {
    v2f pq[2][kTile][64];             /* R1 -> R2 */
    float gp[4][kTile * kTileStride]; /* R2 -> DIV -> W -> SUM: four owners, four buffers (tile & 3) */
    double olaUp[160], olaDown[160];
    float wbin[4][160]; /* binary masks: the only four weight curves there are (none | falling | rising | both) */
};

/* four waves; the caller has excluded L < 320 (no mask frame fits) */
__device__ __forceinline__ void resynth_bwd_body(const ResynthArgs &a, BwdLds &S, int role, int lane, int u,
                                                 long long off, long long L, TilePrio &tp)
{
    v2f(*pq)[kTile][64] = S.pq;
    float(*gp)[kTile * kTileStride] = S.gp;
    double *olaUp = S.olaUp, *olaDown = S.olaDown;
    const long long ntile = (L + kTile - 1) / kTile, niter = ntile + 4;
    for (int i = threadIdx.x; i < 160; i += 256) {
        const double up = a.tables->olaUp[i], down = a.tables->olaDown[i];
        olaUp[i] = up;
        olaDown[i] = down;
        /* with mask values in {0, 1} the weight of a sample is one of four floats that depend on its
         * position in the hop only: (float)(0 + down*1), (float)(0 + up*1), (float)((double)(float)down + up*1) */
        const float wd = (float)(down * 1.0);
        S.wbin[0][i] = 0.0f;
        S.wbin[1][i] = wd;
        S.wbin[2][i] = (float)((double)0.0f + up * 1.0);
        S.wbin[3][i] = (float)((double)wd + up * 1.0);
    }
    tile_sync(tp);

    RS_T_DECL;
    if (role == 0) {
        /* second pass over the time-reversed signal (extractwav.cpp:88), stages 0-1.  Tile tb of
         * this pass is tile ntile-1-tb of the analysis pass read back to front; the steps of the
         * first one that lie past L are zeros (leading zeros leave the all-zero state untouched).
         * HBM latency: the kMacro tiles of the NEXT macro step are requested before the current
         * ones are worked on, one 1 KB request per quarter tile, so a request has kMacro tile
         * periods (~4000 cycles) to complete before it is waited for. */
        const float *src = inter_base(a.inter, off, u) + lane * 4;
        const GtCoef C = gt_coef(a.tables->f1[lane], a.tables->f2[lane]);
        GtLo s = {};
        float4 cur[kMacro][4], nxt[kMacro][4];
        auto request = [&](float4(&b)[kMacro][4], long long macro) {
#pragma unroll
            for (int m = 0; m < kMacro; ++m) {
                long long tf = ntile - 1 - (macro * kMacro + m);
                tf = (tf > 0) ? tf : 0; /* past the end of the pass: harmless re-read of tile 0 */
                const float *row = src + tf * (kTile * 64);
#pragma unroll
                for (int k = 0; k < 4; ++k) b[m][k] = *reinterpret_cast<const float4 *>(row + k * 256);
            }
        };
        request(cur, 0);
        { /* the steps of the very first tile that lie past L are the zero padding */
            const long long top = (ntile - 1) * kTile; /* sample index of stored step 0 of that tile */
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cur[0][k].x = (top + 4 * k + 0 < L) ? cur[0][k].x : 0.0f;
                cur[0][k].y = (top + 4 * k + 1 < L) ? cur[0][k].y : 0.0f;
                cur[0][k].z = (top + 4 * k + 2 < L) ? cur[0][k].z : 0.0f;
                cur[0][k].w = (top + 4 * k + 3 < L) ? cur[0][k].w : 0.0f;
            }
        }
        for (long long j0 = 0; j0 < niter; j0 += kMacro) {
            request(nxt, j0 / kMacro + 1);
#pragma unroll
            for (int m = 0; m < kMacro; ++m) {
                const long long j = j0 + m;
                if (j >= niter) break;
                RS_T_BEGIN;
                if (j < ntile) {
                    v2f(*o)[64] = pq[j & 1];
                    /* step t of this pass = step 15-t of the stored tile: quarters 3..0, each w,z,y,x */
#pragma unroll
                    for (int k = 3; k >= 0; --k) {
                        const v2f lo2 = {cur[m][k].x, cur[m][k].y}, hi2 = {cur[m][k].z, cur[m][k].w};
                        const int t = 4 * (3 - k);
                        o[t + 0][lane] = gt_step_lo(s, hi2.yy, C);
                        o[t + 1][lane] = gt_step_lo(s, hi2.xx, C);
                        o[t + 2][lane] = gt_step_lo(s, lo2.yy, C);
                        o[t + 3][lane] = gt_step_lo(s, lo2.xx, C);
                    }
                }
                RS_T_MID;
                tile_sync(tp);
                RS_T_END;
            }
#pragma unroll
            for (int m = 0; m < kMacro; ++m)
#pragma unroll
                for (int k = 0; k < 4; ++k) cur[m][k] = nxt[m][k];
        }
        RS_T_FLUSH(6);
    } else if (role == 1) {
        const GtCoef C = gt_coef(a.tables->f1[lane], a.tables->f2[lane]);
        const float gain = a.tables->gain[lane];
        GtHi s = {};
        for (long long j = 0; j < niter; ++j) {
            RS_T_BEGIN;
            const long long jt = j - 1;
            if (jt >= 0 && jt < ntile) {
                const v2f(*i)[64] = pq[jt & 1];
                float *o = gp[jt & 3] + lane;
#pragma unroll
                for (int t = 0; t < kTile; ++t) o[t * kTileStride] = gt_step_hi(s, i[t][lane], C, gain);
            }
            RS_T_MID;
            tile_sync(tp);
            RS_T_END;
        }
        RS_T_FLUSH(8);
    } else if (role == 2) {
        /* mask rows: (L-320)/160+1, or L/160 with mode bit 1 (1dnn_resynth/extractwav.cpp:67) */
        const long long F = (a.binary & 2) ? L / 160 : (L - 320) / 160 + 1;
        const float *mask = a.mask + a.mask_offsets[u] * 64 + lane;
        const bool binary = (a.binary & 1) != 0;
        /* mask value of row h as the weight code sees it: the IBM variant turns > 0.5 into 1.0 and
         * skips everything else (resyth_64sub_IBM/cpp/extractwav.cpp:97-99); skipped == 0 here.
         * The load itself (mask_raw) is unconditional and one hop ahead of the interpretation
         * (mask_val), so its latency is never waited for inside the tile loop. */
        auto mask_raw = [&](long long h) -> float {
            const long long hc = (h < 0) ? 0 : ((h >= F) ? F - 1 : h);
            return mask[hc * 64];
        };
        auto mask_val = [&](float raw, long long h) -> float {
            if (h < 0 || h >= F) return 0.0f;
            return binary ? ((raw > 0.5f) ? 1.0f : 0.0f) : raw;
        };
        /* output sample m runs backwards from mTop = 16 ntile - 1 (the samples m >= L of the first
         * tile are the zero padding: computed, never stored) through hops of 160: r = m % 160 counts
         * down.  mh / mh1: mask rows of hop h (falling half) and h+1 (rising half); rawPrev: row h-1. */
        const long long mTop = ntile * kTile - 1;
        long long h = mTop / 160;
        int r = (int)(mTop - h * 160);
        float mh = mask_val(mask_raw(h), h), mh1 = mask_val(mask_raw(h + 1), h + 1);
        float rawPrev = mask_raw(h - 1);
        for (long long j = 0; j < niter; ++j) {
            RS_T_BEGIN;
            const long long jt = j - 3;
            if (jt >= 0 && jt < ntile) {
                float *g = gp[jt & 3] + lane;
                float v[kTile]; /* g / midEar (:89-90), left in place by the fourth wave one tile earlier */
#pragma unroll
                for (int t = 0; t < kTile; ++t) v[t] = g[t * kTileStride];
                if (r >= kTile - 1) {
                    /* whole tile inside one hop (9 tiles out of 10): branch-free, 16 independent
                     * steps for the scheduler.  float(double(0.0f) + x) == float(x), so the first
                     * accumulation needs no add. */
                    const double mhD = (double)mh, mh1D = (double)mh1;
                    const bool useH = mh > 0.0f, useH1 = mh1 > 0.0f;
                    if (binary) { /* IBM variant: table look-up instead of the double-precision products */
                        const float *tab = S.wbin[(useH ? 1 : 0) | (useH1 ? 2 : 0)] + r;
#pragma unroll
                        for (int t = 0; t < kTile; ++t) g[t * kTileStride] = tab[-t] * v[t];
                    } else {
#pragma unroll
                        for (int t = 0; t < kTile; ++t) {
                            const float w1 = (float)(olaDown[r - t] * mhD);
                            float w = useH ? w1 : 0.0f;
                            const float w2 = (float)((double)w + olaUp[r - t] * mh1D);
                            w = useH1 ? w2 : w;
                            g[t * kTileStride] = w * v[t]; /* :108-112 term of this channel */
                        }
                    }
                    r -= kTile;
                    if (r < 0) { /* step into hop h-1 */
                        r += 160;
                        h--;
                        mh1 = mh;
                        mh = mask_val(rawPrev, h);
                        rawPrev = mask_raw(h - 1);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < kTile; ++t) {
                        float w = 0.0f; /* :91-107: falling half of frame h, then rising half of h+1 */
                        if (mh > 0.0f) w = (float)((double)w + olaDown[r] * (double)mh);
                        if (mh1 > 0.0f) w = (float)((double)w + olaUp[r] * (double)mh1);
                        g[t * kTileStride] = w * v[t];
                        if (--r < 0) { /* step into hop h-1: the next row was requested a hop ago */
                            r = 159;
                            h--;
                            mh1 = mh;
                            mh = mask_val(rawPrev, h);
                            rawPrev = mask_raw(h - 1);
                        }
                    }
                }
            }
            RS_T_MID;
            tile_sync(tp);
            RS_T_END;
        }
        RS_T_FLUSH(10);
    } else {
        int16_t *out = a.out + off;
        const long long mTop = ntile * kTile - 1;
        const DivConst ear = div_const(a.tables->midEar[lane]);
        for (long long j = 0; j < niter; ++j) {
            RS_T_BEGIN;
            const long long jd = j - 2, jt = j - 4;
            if (jd >= 0 && jd < ntile) { /* lane = channel: reverse[...] / midEar (:89-90) of the tile R2 finished, in place */
                float *g = gp[jd & 3] + lane;
                float gv[kTile], v[kTile];
#pragma unroll
                for (int t = 0; t < kTile; ++t) gv[t] = g[t * kTileStride];
                div_tile16(gv, v, ear);
#pragma unroll
                for (int t = 0; t < kTile; ++t) g[t * kTileStride] = v[t];
            }
            if (jt >= 0 && jt < ntile) {
                if (lane < kTile) { /* channel sum in order 0..63 for step t = lane, (short) cast :120-121 */
                    const long long m = mTop - jt * kTile - lane;
                    const float4 *row = reinterpret_cast<const float4 *>(gp[jt & 3] + lane * kTileStride);
                    float acc = 0.0f;
#pragma unroll
                    for (int c4 = 0; c4 < 16; ++c4) {
                        const float4 p = row[c4];
                        acc += p.x;
                        acc += p.y;
                        acc += p.z;
                        acc += p.w;
                    }
                    if (m < L) out[m] = (int16_t)cast_i16(acc);
                }
            }
            RS_T_MID;
            tile_sync(tp);
            RS_T_END;
        }
        RS_T_FLUSH(12);
    }
}

} // namespace

__global__ __launch_bounds__(256, 4) void resynth_bwd_kernel(ResynthArgs a)
{
    __shared__ BwdLds S;
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const long long L = a.lengths[u];
    if (L < ((a.binary & 2) ? 160 : 320)) return; /* no mask frame fits (wave-uniform exit before any barrier) */
    TilePrio tp = {0, 0.0f};
    resynth_bwd_body(a, S, role, lane, u, a.offsets[u], L, tp);
}

/* Both passes of one utterance in ONE workgroup, back to back: the analysis pass of the long
 * utterances no longer has to drain (and idle most of the chip) before any synthesis pass may
 * start -- a workgroup that finishes its analysis moves straight on, so the two tails overlap with
 * other utterances' work.  The intermediate still goes through HBM (the synthesis pass reads it
 * backwards); the hand-over inside the workgroup is a release / acquire pair at agent scope. */
__global__ __launch_bounds__(256, 4) void resynth_fused_kernel(ResynthArgs a)
{
    __shared__ union U {
        FwdLds f;
        BwdLds b;
        __device__ U() {}
    } S;
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    /* (issue priority by launch ROW, as the NoiseSup kernel had it in round 3: 13.55 / 13.92 against 13.56 / 13.46 ms, no gain; by
     * REMAINING tiles -- TilePrio, round 4 -- 13.36 against 13.60 ms (ratio masks), 12.74 against 13.05 (binary), alternating A/B) */
    const long long off = a.offsets[u], L = a.lengths[u];
    if (L < ((a.binary & 2) ? 160 : 320)) return; /* no mask frame fits */
    /* both passes' tile barriers of this utterance against those of the batch's longest (block 0's: longest first) */
    TilePrio tp = {0, 0.0f};
    if (a.order && gridDim.x > 1) {
        const long long Lmax = a.lengths[a.order[0]];
        tp.left = 2 * ((L + kTile - 1) / kTile) + 7;
        tp.scale = 32.0f / (float)(2 * ((Lmax + kTile - 1) / kTile) + 7);
    }
#ifdef SEA_RS_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_rs_wg[4 * blockIdx.x] = (unsigned)wall_clock64();
        g_rs_wg[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        g_rs_wg[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    }
#endif
    resynth_fwd_body(a, S.f, role, lane, u, off, L, tp);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    resynth_bwd_body(a, S.b, role, lane, u, off, L, tp);
#ifdef SEA_RS_TIMING
    if (threadIdx.x == 192 && blockIdx.x < 4096) g_rs_wg[4 * blockIdx.x + 1] = (unsigned)wall_clock64(); /* the SUM wave ends last */
#endif
}

/* ---- subbband() -- the analysis half on its own --------------------------------
 * gammatone -> Meddis hair cell -> (short) cast -> 64 int16 streams per utterance
 * (enhancement_extract_test/cpp/extractwav.cpp:40-101; hairCell resyth_64sub_ori/cpp/extractwav.cpp:
 * 212-257, constants HuWang.h:36-44).  Output is 64x the input: 128 B written per input sample, the
 * one genuinely HBM-write-heavy kernel of the path.  Workgroup = one utterance = six waves,
 * lane = channel, 16-sample tiles:
 *   R1, R2  the two halves of the gammatone cascade (as in the resynthesis kernels)
 *   K0, K1  the hair cell's input-only permeability kt (a double division per sample), even / odd steps
 *   HC      the q/c/w recurrence, output hdt*c truncated to int16 into a padded LDS tile
 *   W       transposes the tile (lane = 16 samples x 4 channels) and streams 32-byte runs of each
 *           channel's row; consecutive tiles complete the 128-byte lines in L2. */
namespace {

struct HairCell { /* Meddis 1988 constants as the reference's C++ evaluates them (float dt, double literals) */
    float ymdt, xdt, ydt, lplusrdt, rdt, gdt, hdt;
    float q, c, w;
};

__device__ __forceinline__ void haircell_init(HairCell &h)
{
    const double Y = 5.05, G = 2000.0, Lc = 2500.0, R = 6580.0, X = 66.31, A = 3.0, B = 300.0, H = 48000.0, M = 1.0;
    const float dt = 1 / (float)16000;
    h.ymdt = (float)(Y * M * dt);
    h.xdt = (float)(X * dt);
    h.ydt = (float)(Y * dt);
    h.lplusrdt = (float)((Lc + R) * dt);
    h.rdt = (float)(R * dt);
    h.gdt = (float)(G * dt);
    h.hdt = (float)H;
    const float kt = (float)(G * A / (A + B));
    h.c = (float)(M * Y * kt / (Lc * kt + Y * (Lc + R)));
    h.q = (float)(h.c * (Lc + R) / kt);
    h.w = (float)(h.c * R / X);
}

/* permeability of one sample (extractwav.cpp:237): depends on the input only */
__device__ __forceinline__ float haircell_kt(const HairCell &h, float in)
{
    const double s = (double)in + 3.0;
    return (s > 0.0) ? (float)((double)h.gdt * s / (s + 300.0)) : 0.0f;
}

/* the recurrence (extractwav.cpp:239-255); returns output[n] = hdt * c */
__device__ __forceinline__ float haircell_step(HairCell &h, float kt)
{
    const float replenish = ((double)h.q < 1.0) ? (h.ymdt - h.ydt * h.q) : 0.0f;
    const float eject = kt * h.q;
    const float reuptakeandloss = h.lplusrdt * h.c;
    const float reuptake = h.rdt * h.c;
    const float reprocess = h.xdt * h.w;
    h.q = h.q + replenish - eject + reprocess;
    if (h.q < 0.0f) h.q = 0.0f;
    h.c = h.c + eject - reuptakeandloss;
    if (h.c < 0.0f) h.c = 0.0f;
    h.w = h.w + reuptake - reprocess;
    if (h.w < 0.0f) h.w = 0.0f;
    return h.hdt * h.c;
}

/* the same step with fewer vector instructions on the recurrence's wave (round 4): the pair (c, w) through packed operations --
 * each half rounded like the scalar instruction -- and the three clamps as v_max_f32 against +0 instead of compare + select.
 * The clamp differs from `if (x < 0) x = 0` only for NaN and for -0.0, neither of which the recurrence can produce: every input is
 * finite (int16 audio through the gammatone cascade, kt in [0, gdt)), the state stays >= +0, and a sum or difference of finite
 * values that is zero is +0 in round-to-nearest unless both operands are -0.  15 instead of 21 instructions per sample. */
__device__ __forceinline__ void haircell_step_lean(HairCell &h, float kt)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    const float replenish = (h.q < 1.0f) ? (h.ymdt - h.ydt * h.q) : 0.0f; /* (double)q < 1.0 <=> q < 1.0f: the conversion is exact */
    const float eject = kt * h.q;
    const v2 cc = {h.c, h.c};
    const v2 rl = v2{h.lplusrdt, h.rdt} * cc;      /* reuptakeandloss, reuptake */
    const float reprocess = h.xdt * h.w;
    const float q = h.q + replenish - eject + reprocess;
    v2 cw = v2{h.c, h.w} + v2{eject, rl.y};        /* c + eject, w + reuptake */
    cw = cw - v2{rl.x, reprocess};                 /* - reuptakeandloss, - reprocess */
    h.q = __builtin_fmaxf(q, 0.0f);
    h.c = __builtin_fmaxf(cw.x, 0.0f);
    h.w = __builtin_fmaxf(cw.y, 0.0f);
}
#ifndef SEA_SB_HC_LEAN
#define SEA_SB_HC_LEAN 1
#endif

constexpr int kSbStride = 66; /* int16 per tile row: 64 channels + 2 pad (33 words: conflict-free) */
/* SEA_SB_CAST_IN_W (round 4): the hair cell's wave -- the longest role of subbband() by half (2850 clk per 16-sample tile against
 * 1900 for the next) -- leaves its state c of every step where its input kt stood (the permeability tile, now a ring of three
 * float tiles of row stride 66) and the transposing wave W, idle four fifths of a tile, takes hdt * c, the truncating cast and the
 * store: four of the 29 vector instructions per sample leave the recurrence's wave.  Same operations on the same values. */
#ifndef SEA_SB_CAST_IN_W
#define SEA_SB_CAST_IN_W 1
#endif
constexpr int kKtStride = SEA_SB_CAST_IN_W ? 66 : 64, kKtBufs = SEA_SB_CAST_IN_W ? 3 : 2;

} // namespace

__global__ __launch_bounds__(384) void subband_kernel(SubbandArgs a)
{
    __shared__ RsLds S;
    __shared__ __attribute__((aligned(16))) float xs[kTile];
    __shared__ __attribute__((aligned(16))) float ktile[kKtBufs][kTile][kKtStride];
    __shared__ __attribute__((aligned(16))) short otile[SEA_SB_CAST_IN_W ? 1 : 2][SEA_SB_CAST_IN_W ? 8 : kTile * kSbStride];
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const long long off = a.offsets[u], L = a.lengths[u];
    const long long ntile = (L + kTile - 1) / kTile, niter = ntile + 4;
    /* (issue priority by remaining tiles, as the fused resynthesis kernel has it: 14.0-14.1 against 13.6-13.7 ms here -- six-wave
     * workgroups, two per CU: off) */
    RS_T_DECL;
    if (role == 0) {
        const int16_t *in = a.in + off;
        const GtCoef C = gt_coef(a.tables->f1[lane], a.tables->f2[lane]);
        GtLo s = {};
        InFeed feed;
        feed.start(in, L, lane);
        for (long long j = 0; j < niter; ++j) {
            RS_T_BEGIN;
            if (j < ntile) {
                feed.tile(in, L, j, lane, xs);
                wave_sync();
                v2f(*o)[64] = S.pq[j & 1];
#pragma unroll
                for (int t = 0; t < kTile; t += 2) {
                    const v2f x2 = *reinterpret_cast<const v2f *>(&xs[t]);
                    o[t][lane] = gt_step_lo(s, x2.xx, C);
                    o[t + 1][lane] = gt_step_lo(s, x2.yy, C);
                }
                wave_sync();
            }
            RS_T_MID;
            tile_sync();
            RS_T_END;
        }
        RS_T_FLUSH(16);
    } else if (role == 1) {
        const GtCoef C = gt_coef(a.tables->f1[lane], a.tables->f2[lane]);
        const float gain = a.tables->gain[lane];
        GtHi s = {};
        for (long long j = 0; j < niter; ++j) {
            RS_T_BEGIN;
            const long long jt = j - 1;
            if (jt >= 0 && jt < ntile) {
                const v2f(*i)[64] = S.pq[jt & 1];
                float(*o)[64] = S.g[jt & 1];
#pragma unroll
                for (int t = 0; t < kTile; ++t) o[t][lane] = gt_step_hi(s, i[t][lane], C, gain);
            }
            RS_T_MID;
            tile_sync();
            RS_T_END;
        }
        RS_T_FLUSH(18);
    } else if (role == 2 || role == 3) {
        /* K (two waves, even / odd steps): the hair cell's permeability, input-only; a correctly rounded
         * double division per sample is ~35 dependent instructions, the longest stage by far */
        HairCell h;
        haircell_init(h);
        const int par = role - 2;
        for (long long j = 0; j < niter; ++j) {
            RS_T_BEGIN;
            const long long jt = j - 2;
            if (jt >= 0 && jt < ntile) {
                const float(*g)[64] = S.g[jt & 1];
                float(*o)[kKtStride] = ktile[jt % kKtBufs];
#pragma unroll
                for (int t = 0; t < kTile; t += 2) o[t + par][lane] = haircell_kt(h, g[t + par][lane]);
            }
            RS_T_MID;
            tile_sync();
            RS_T_END;
        }
        RS_T_FLUSH(20);
    } else if (role == 4) {
        /* HC: the q/c/w recurrence and the (short) cast of hOut (extractwav.cpp:85-88) */
        HairCell h;
        haircell_init(h);
        for (long long j = 0; j < niter; ++j) {
            RS_T_BEGIN;
            const long long jt = j - 3;
            if (jt >= 0 && jt < ntile) {
                float(*k)[kKtStride] = ktile[jt % kKtBufs];
                short *o = otile[SEA_SB_CAST_IN_W ? 0 : (jt & 1)];
#pragma unroll
                for (int t = 0; t < kTile; ++t) {
                    if (SEA_SB_CAST_IN_W && SEA_SB_HC_LEAN) {
                        haircell_step_lean(h, k[t][lane]);
                        k[t][lane] = h.c;
                    } else {
                        const float v = haircell_step(h, k[t][lane]);
                        if (SEA_SB_CAST_IN_W) k[t][lane] = h.c; /* W takes hdt * c and the cast */
                        else o[t * kSbStride + lane] = (short)cast_i16(v);
                    }
                }
            }
            RS_T_MID;
            tile_sync();
            RS_T_END;
        }
        RS_T_FLUSH(22);
    } else {
        /* W: lane = sample t (0..15) of channel group cg (0..3): 16 passes cover the 64 channels */
        int16_t *out = a.out + off * 64;
        const long long Lp = (L + 7) & ~7LL; /* row pitch of this utterance's [64][Lp] block */
        const int t = lane & 15, cg = lane >> 4;
        HairCell hw;
        haircell_init(hw);
        const float hdtW = hw.hdt;
        for (long long j = 0; j < niter; ++j) {
            RS_T_BEGIN;
            const long long jt = j - 4;
            if (jt >= 0 && jt < ntile) {
                const long long n = jt * kTile + t;
                const short *o = otile[SEA_SB_CAST_IN_W ? 0 : (jt & 1)] + (SEA_SB_CAST_IN_W ? 0 : t * kSbStride);
                const float *cf = &ktile[jt % kKtBufs][t][0];
                if (n < L) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int c = 4 * k + cg;
                        if (SEA_SB_CAST_IN_W) out[c * Lp + n] = (short)cast_i16(hdtW * cf[c]); /* extractwav.cpp:255, :85-88 */
                        else out[c * Lp + n] = o[c];
                    }
                }
            }
            RS_T_MID;
            tile_sync();
            RS_T_END;
        }
        RS_T_FLUSH(24);
    }
}

/* sea_selftest_div: every float pattern a inside the fast-division domain, every channel's divisor:
 * div_fast / div_fast2 against a / d.  mismatches[0] = count of differing results, [1] = patterns
 * tested per channel (domain size), for the host to report. */
__global__ __launch_bounds__(256) void selftest_div_kernel(const sea_gt_tables *t, unsigned long long *mismatches)
{
    unsigned long long bad = 0, tested = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (int ch = 0; ch < 64; ++ch) {
        const float d = t->midEar[ch];
        const DivConst c = div_const(d);
        for (unsigned long long v = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; v < (1ull << 32); v += stride) {
            const float a = __uint_as_float((unsigned)v);
            if (!div_in_domain(a)) continue;
            const float want = a / d;
            const float got = div_fast(a, c);
            const v2f got2 = div_fast2(v2f{a, -a}, c);
            bad += (__float_as_uint(got) != __float_as_uint(want)) ? 1 : 0;
            bad += (__float_as_uint(got2.x) != __float_as_uint(want)) ? 1 : 0;
            bad += (__float_as_uint(got2.y) != __float_as_uint(-want)) ? 1 : 0;
            if (ch == 0) tested++;
        }
    }
    if (bad) atomicAdd(&mismatches[0], bad);
    if (tested) atomicAdd(&mismatches[1], tested);
}

} // namespace sea

#ifdef SEA_RS_TIMING
extern "C" int sea_debug_rs_wg(unsigned *out, int n_wg)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sea::g_rs_wg), (size_t)n_wg * 4 * sizeof(unsigned));
}
extern "C" int sea_debug_rs_timing(unsigned long long *out16)
{
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(sea::g_rs_timing), 32 * sizeof(unsigned long long));
}
#endif
