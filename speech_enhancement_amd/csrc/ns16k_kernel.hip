/*
 * ns16k_kernel.hip -- the 16 k-native NoiseSup variant behind the reference's batch plug-in symbols:
 *     function/20141106_speech_enhancement/aurora_etsi/NoiseSup.cpp:1140-1407   etsi_denoise_mapping_func_Wiener
 *     function/20141106_speech_enhancement/aurora_etsi/NoiseSup.h:36-53         hop 160, window 480 at offset 80 of a
 *                                                                               640-sample stage buffer, 129 spectral
 *                                                                               values, NS_FFT_LENGTH 512 / ORDER 8
 *     function/20141106_speech_enhancement/aurora_etsi/MelProc.cpp:119-135,556-576   DoGamma, DoGammaIDCT
 *     function/20141106_speech_enhancement/aurora_etsi/rfft.cpp:46-181          rfft (x, 512, 8)
 *
 * One 64-lane wavefront per stream, the whole per-stream state in LDS between the frames of a push and in a
 * kNs16StateFloats-float blob in HBM between pushes.  This is a SIDE path (the reference's hot path is etsi/'s 8 kHz
 * framing, ns_pipe_kernel.hip): written for exactness and clarity, every division and square root on the exact path, the
 * transform as a table of independent butterflies per pass (sea_ns16k_tables).  The scalar recursions (VAD, gain
 * factorisation, SpeechQ*) are the ones of ns_core.h, which the two variants share word for word
 * (diff of etsi/cpp/NoiseSup.c against aurora_etsi/NoiseSup.cpp: sizes, the gammatone windows, the frame gate).
 * Deviation stated in oracle/ns16k_oracle.c: DoGainFact_IBM's log10 is evaluated in double, as in the C tree.
 */
#include "ns_core.h"

namespace sea {

namespace {

constexpr int kHop = SEA16_HOP, kBuf = SEA16_BUF, kIn = SEA16_BUF - SEA16_HOP /* NS_DATA_IN_BUFFER */, kSpec = SEA16_NSPEC;
constexpr int kSpecPad = 132;

struct __attribute__((aligned(16))) Ns16Lds {
    float buf[2][kBuf];       /* First / SecondStageInFloatBuffer */
    float bins[6][kSpecPad];  /* noiseSE1/2, denSigSE1/2, the other slot of PSDMeanBuffer1/2 */
    float work[SEA16_NFFT];   /* transform workspace */
    float W[kSpecPad];        /* Wiener gains of the 129 values */
    float sq[kHop];           /* squares (frame gate, VAD), then the DC filter's differences */
    float outb[kHop];         /* second-stage output frame */
    float gam[32];            /* the 25 window outputs */
    float fir[20];            /* 17 taps */
    float gammaT[SEA16_GLEN][SEA16_NGAM];
};

/* rfft (x, 512, 8) on L.work, elements already at their digit-reversed places */
__device__ __forceinline__ void ns16_fft(float *x, const sea_ns16k_tables *t, int lane)
{
#pragma unroll 1
    for (int pass = 0; pass < SEA16_FFT_PASSES; ++pass) {
        const int cnt = (int)t->fftCount[pass];
        const int n4 = (pass == 0) ? 0 : (1 << (pass - 1)), n8 = n4 >> 1; /* n2 = 2^(pass+1) */
#pragma unroll 1
        for (int r = lane; r < cnt; r += kLanes) {
            const unsigned it = t->fftItem[pass][r];
            const int kind = (int)(it >> 24), j = (int)((it >> 16) & 0xffu), i = (int)(it & 0xffffu);
            if (kind == SEA16_BF_LEN2) { /* rfft.cpp:86-91 */
                const float a0 = x[i], a1 = x[i + 1];
                x[i] = a0 + a1;
                x[i + 1] = a0 - a1;
            } else if (kind == SEA16_BF_PLAIN) { /* :113-120 */
                const int i1 = i, i3 = i + 2 * n4, i4 = i + 3 * n4;
                const float x1 = x[i1], x3 = x[i3], x4 = x[i4];
                const float t1 = x4 + x3;
                x[i4] = x4 - x3;
                x[i3] = x1 - t1;
                x[i1] = x1 + t1;
            } else if (kind == SEA16_BF_PI4) { /* :122-133: the sums are divided by sqrt 2 in double */
                const int i1 = i + n8, i2 = i1 + n4, i3 = i2 + n4, i4 = i3 + n4;
                const float x1 = x[i1], x2 = x[i2], x3 = x[i3], x4 = x[i4];
                const float t1 = (float)((double)(x3 + x4) / 1.41421356237309504880);
                const float t2 = (float)((double)(x3 - x4) / 1.41421356237309504880);
                x[i4] = x2 - t1;
                x[i3] = -x2 - t1;
                x[i2] = x1 - t2;
                x[i1] = x1 + t2;
            } else { /* :139-176 */
                const float cc1 = t->fftTw[pass][j][0], ss1 = t->fftTw[pass][j][1], cc3 = t->fftTw[pass][j][2],
                            ss3 = t->fftTw[pass][j][3];
                const int i1 = i + j, i2 = i1 + n4, i3 = i2 + n4, i4 = i3 + n4;
                const int i5 = i + n4 - j, i6 = i5 + n4, i7 = i6 + n4, i8 = i7 + n4;
                const float x1 = x[i1], x2 = x[i2], x3 = x[i3], x4 = x[i4], x5 = x[i5], x6 = x[i6], x7 = x[i7], x8 = x[i8];
                float t1 = x3 * cc1 + x7 * ss1;
                float t2 = x7 * cc1 - x3 * ss1;
                float t3 = x4 * cc3 + x8 * ss3;
                float t4 = x8 * cc3 - x4 * ss3;
                const float t5 = t1 + t3, t6 = t2 + t4;
                t3 = t1 - t3;
                t4 = t2 - t4;
                x[i3] = t6 - x6;
                x[i8] = x6 + t6;
                x[i7] = -x2 - t3;
                x[i4] = x2 - t3;
                x[i6] = x1 - t5;
                x[i1] = x1 + t5;
                x[i5] = x5 - t4;
                x[i2] = x5 + t4;
            }
        }
        wave_sync(); /* the butterflies of one pass touch disjoint elements; passes are ordered */
    }
}

/* DCOffsetFil over one 160-sample frame (NoiseSup.cpp:168-184) in the float-FMA form of ns_core.h's dc_filter: the
 * exactness condition is checked for every sample afterwards, the frame redone in double where it fails */
__device__ __forceinline__ void ns16_dc(const float *dif, float *out, float &yState, int lane)
{
    const float y0 = yState;
    float y = y0;
#pragma unroll 4
    for (int n = 0; n < kHop; n += 4) {
        const float4 d = *reinterpret_cast<const float4 *>(&dif[n]);
        float4 o;
        y = __fmaf_rn(0.9990234375f, y, d.x);
        o.x = y;
        y = __fmaf_rn(0.9990234375f, y, d.y);
        o.y = y;
        y = __fmaf_rn(0.9990234375f, y, d.z);
        o.z = y;
        y = __fmaf_rn(0.9990234375f, y, d.w);
        o.w = y;
        *reinterpret_cast<float4 *>(&out[n]) = o;
    }
    wave_sync();
    bool unsafe = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int n = lane + kLanes * k;
        if (n < kHop) unsafe |= !dc_step_ok(dif[n], n == 0 ? y0 : out[n - 1]);
    }
    if (__ballot(unsafe) != 0ull) {
        wave_sync();
        y = y0;
        for (int n = 0; n < kHop; ++n) {
            y = (float)__fma_rn(0.9990234375, (double)y, (double)dif[n]);
            out[n] = y;
        }
    }
    yState = y;
    wave_sync();
}

/* one stage of one frame (:1207-1366) */
template <int ST>
__device__ __forceinline__ void ns16_stage(Ns16Lds &L, NsRegs &s, NsFd &fd, const sea_ns16k_tables *t, const float (&idct)[SEA16_NGAM],
                                           float irWin, int lane, int &fdBits, float *wienerRow)
{
    float *buf = L.buf[ST];
    /* window + zero padding (:209-222) straight to the digit-reversed places */
#pragma unroll
    for (int k = 0; k < SEA16_NFFT / kLanes; ++k) {
        const int i = lane + kLanes * k;
        const float v = (i < SEA16_WIN) ? buf[SEA16_AWIN + i] * t->sigWindow[i] : 0.0f;
        L.work[t->rev[i]] = v;
    }
    wave_sync();
    ns16_fft(L.work, t, lane);

    /* FFTtoPSD (:240-261), PSDMean (:280-294) and FilterCalc (:440-553) per spectral value: b = lane, lane + 64, 128 */
    {
        int nb = s.nbFrame[ST];
        if (nb < 2147483647) nb++;
        s.nbFrame[ST] = nb;
    }
    if (ST == 0) { /* _VAD_ (:350-421) on curFrame = buf[160..319] */
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = lane + kLanes * k;
            if (i < kHop) {
                const float x = buf[kHop + i];
                L.sq[i] = x * x;
            }
        }
        wave_sync();
        vad_update(s, vad_frame_energy(serial_sum<kHop>(L.sq, 64.0f)));
    }
    const int nb16 = (int)(short)s.nbFrame[ST];
    const float *x = L.work;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int b = lane + kLanes * k;
        const bool act = b < kSpec;
        const int bb = act ? b : (kSpec - 1);
        float nSig;
        if (bb == kSpec - 1) {
            nSig = x[256] * x[256];
        } else {
            const int j0 = 2 * bb, j1 = 2 * bb + 1;
            const float p0 = (bb == 0) ? x[0] * x[0] : (x[j0] * x[j0] + x[SEA16_NFFT - j0] * x[SEA16_NFFT - j0]);
            const float p1 = x[j1] * x[j1] + x[SEA16_NFFT - j1] * x[SEA16_NFFT - j1];
            nSig = (float)((double)(p0 + p1) / 2.0);
        }
        const float P = (L.bins[4 + ST][bb] + nSig) / 2.0f;
        float noise = L.bins[ST][bb], den = L.bins[2 + ST][bb];
        const float W = filter_bin<ST, false>(P, nSig, noise, den, nb16, s.flagVAD, t->eps);
        if (act) {
            L.bins[4 + ST][b] = nSig;
            L.bins[ST][b] = noise;
            L.bins[2 + ST][b] = den;
            L.W[b] = W;
        }
    }
    wave_sync();
    if (ST == 0) fdBits = fd_var<SEA16_NFFT / 4>(fd, L.W, nb16); /* SpeechQVar (:852-893) on the first 128 gains */

    /* DoGamma (MelProc.cpp:119-135): window c = lane over gains 0..127, in order */
    {
        float sum = 0.0f;
        const int c = (lane < SEA16_NGAM) ? lane : 0;
#pragma unroll 8
        for (int i = 0; i < SEA16_GLEN; ++i) sum += L.W[i] * L.gammaT[i][c];
        wave_sync(); /* all 128 gains read before the outputs overwrite nothing here: gam is its own array */
        if (lane < SEA16_NGAM) L.gam[lane] = sum;
    }
    wave_sync();
    if (ST == 0) { /* :1301-1311, :1362-1365 */
        fdBits |= fd_spec_mel(fd, L.gam, nb16) << 1;
        fdBits |= (s.nbSpeech > 4) ? 8 : 0;
    }
    /* DoGainFact_IBM (:634-698) */
    float g = L.gam[(lane < SEA16_NGAM) ? lane : 0];
    if (ST == 0) {
        const float total = serial_sum<kSpec>(L.bins[2], 0.0f);
        s.denEn0 = s.denEn1;
        s.denEn1 = s.denEn2;
        s.denEn2 = total;
    } else {
        gain_fact_update(s, serial_sum<kSpec>(L.bins[1], 0.0f));
        g = (float)((double)(s.alfaGF * g) + (1.0 - (double)s.alfaGF) * 1.0);
        if (wienerRow && lane < SEA16_NGAM) wienerRow[lane] = g; /* the line func_Wiener prints (:1319-1328) */
        wave_sync();
        if (lane < SEA16_NGAM) L.gam[lane] = g;
        wave_sync();
    }
    /* DoGammaIDCT (MelProc.cpp:556-576), taps t = lane = 0..8, + DoFilterWindowing (:716-725) */
    {
        float h = 0.0f;
#pragma unroll
        for (int f = 0; f < SEA16_NGAM; ++f) h += L.gam[f] * idct[f];
        const float tap = h * irWin;
        if (lane <= 8) {
            L.fir[8 + lane] = tap;
            L.fir[8 - lane] = tap;
        }
    }
    wave_sync();
    /* ApplyWF (cur, prv, filterIR, out, 160, 8) (:317-331): predata[160 - j + i] is cur[i - j] in the contiguous stage
     * buffer, so both loops are one sum over j = -8..8 in that order */
    float *dst = (ST == 0) ? (L.buf[1] + kIn) : L.outb;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = lane + kLanes * k;
        if (i < kHop) {
            float acc = 0.0f;
#pragma unroll
            for (int j = -8; j <= 8; ++j) acc += L.fir[j + 8] * buf[kHop + i - j];
            dst[i] = acc;
        }
    }
    wave_sync();
}

/* slide a stage buffer by one hop (:1372-1390) */
__device__ __forceinline__ void ns16_slide(float *buf, int lane)
{
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = lane + kLanes * k;
        r[k] = (i < kIn) ? buf[i + kHop] : 0.0f;
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = lane + kLanes * k;
        if (i < kIn) buf[i] = r[k];
    }
}

constexpr int kBlobBins = 2 * kBuf, kBlobScal16 = kBlobBins + 6 * kSpecPad;
static_assert(kBlobScal16 + 32 == kNs16StateFloats, "state blob layout");

} // namespace

__global__ void __launch_bounds__(64) ns16k_stream_kernel(Ns16StreamArgs a)
{
    __shared__ Ns16Lds L;
    const int lane = threadIdx.x, b = blockIdx.x;
    const sea_ns16k_tables *t = a.tables;
    NsRegs s;
    NsFd fd;
    float *blob = a.state + (size_t)b * kNs16StateFloats;

    for (int i = lane; i < SEA16_GLEN * SEA16_NGAM; i += kLanes) (&L.gammaT[0][0])[i] = (&t->gammaT[0][0])[i];
    float idct[SEA16_NGAM];
#pragma unroll
    for (int f = 0; f < SEA16_NGAM; ++f) idct[f] = t->idctT[f][lane & 15];
    const float irWin = t->irWin[lane & 15];

    if (a.reset) { /* etsi_denoise_mapping_thread_init (:937-1083) */
        regs_init(s, t->eps);
        fd_init(fd);
        for (int i = lane; i < 2 * kBuf; i += kLanes) (&L.buf[0][0])[i] = 0.0f;
        for (int i = lane; i < 6 * kSpecPad; i += kLanes) (&L.bins[0][0])[i] = (i < 2 * kSpecPad) ? t->eps : 0.0f;
    } else {
        for (int i = lane; i < 2 * kBuf; i += kLanes) (&L.buf[0][0])[i] = blob[i];
        for (int i = lane; i < 6 * kSpecPad; i += kLanes) (&L.bins[0][0])[i] = blob[kBlobBins + i];
        const float *q = blob + kBlobScal16;
        const int *qi = reinterpret_cast<const int *>(q + 16);
        regs_init(s, t->eps);
        s.dcX = q[0]; s.dcY = q[1]; s.denEn0 = q[2]; s.denEn1 = q[3]; s.denEn2 = q[4];
        s.lowSNRtrack = q[5]; s.alfaGF = q[6]; s.meanEn = q[7];
        fd.melMean = q[8]; fd.varMean = q[9]; fd.accTest = q[10]; fd.specMean = q[11];
        fd.mel0 = q[12]; fd.specValues = q[13]; fd.speechInVADQ = q[14];
        s.nbFrame[0] = qi[0]; s.nbFrame[1] = qi[1]; s.flagVAD = qi[2]; s.hangOver = qi[3];
        s.nbSpeech = qi[4]; s.nIn1 = qi[5]; s.nIn2 = qi[6]; s.nOut2 = qi[7];
    }
    wave_sync();

    const float *in = a.in + (size_t)b * a.nframes * kHop;
    float *out = a.out + (size_t)b * a.nframes * kHop;
    for (int f = 0; f < a.nframes; ++f, in += kHop, out += kHop) {
        const size_t rec = (size_t)b * a.nframes + f;
        /* the frame gate (:1160-1171): float sum of squares in sample order, truncated to int */
        float xin[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = lane + kLanes * k;
            xin[k] = (i < kHop) ? in[i] : 0.0f;
            if (i < kHop) L.sq[i] = xin[k] * xin[k];
        }
        wave_sync();
        const float check = uniform_f(serial_sum<kHop>(L.sq, 0.0f));
        int produced = 0, counter = 0, fdBits = 0;
        if (!(check < 1.0f)) { /* (int)check != 0; NaN and overflow convert to INT_MIN on the reference's x86 */
            wave_sync();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = lane + kLanes * k;
                if (i < kHop) L.buf[0][kIn + i] = xin[k];
            }
            wave_sync();
            s.nIn1++;
            if (s.nIn1 - s.nIn2 > 2) { /* :1212 */
                ns16_stage<0>(L, s, fd, t, idct, irWin, lane, fdBits, nullptr);
                s.nIn2++;
                counter = s.nbFrame[0];
            }
            if (s.nIn2 - s.nOut2 > 2) { /* :1230 */
                int unused = 0;
                ns16_stage<1>(L, s, fd, t, idct, irWin, lane, unused, a.wiener ? a.wiener + rec * SEA16_NGAM : nullptr);
                s.nOut2++;
                produced = 1;
            }
            ns16_slide(L.buf[0], lane);
            if (s.nIn2) ns16_slide(L.buf[1], lane);
            wave_sync();
            if (s.nOut2 > 0) { /* DCOffsetFil on the frame just written (:1392-1395) */
                float d[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int i = lane + kLanes * k;
                    d[k] = (i < kHop) ? (L.outb[i] - ((i == 0) ? s.dcX : L.outb[i - 1])) : 0.0f;
                }
                s.dcX = L.outb[kHop - 1];
                wave_sync();
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int i = lane + kLanes * k;
                    if (i < kHop) L.sq[i] = d[k];
                }
                wave_sync();
                ns16_dc(L.sq, L.outb, s.dcY, lane);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int i = lane + kLanes * k;
                    if (i < kHop) out[i] = L.outb[i];
                }
            }
        }
        if (lane == 0) {
            a.produced[rec] = produced;
            if (a.flags) a.flags[rec] = (unsigned char)fdBits;
            if (a.frame_counter) a.frame_counter[rec] = counter;
        }
        wave_sync();
    }

    for (int i = lane; i < 2 * kBuf; i += kLanes) blob[i] = (&L.buf[0][0])[i];
    for (int i = lane; i < 6 * kSpecPad; i += kLanes) blob[kBlobBins + i] = (&L.bins[0][0])[i];
    if (lane == 0) {
        float *q = blob + kBlobScal16;
        int *qi = reinterpret_cast<int *>(q + 16);
        q[0] = s.dcX; q[1] = s.dcY; q[2] = s.denEn0; q[3] = s.denEn1; q[4] = s.denEn2;
        q[5] = s.lowSNRtrack; q[6] = s.alfaGF; q[7] = s.meanEn;
        q[8] = fd.melMean; q[9] = fd.varMean; q[10] = fd.accTest; q[11] = fd.specMean;
        q[12] = fd.mel0; q[13] = fd.specValues; q[14] = fd.speechInVADQ;
        qi[0] = s.nbFrame[0]; qi[1] = s.nbFrame[1]; qi[2] = s.flagVAD; qi[3] = s.hangOver;
        qi[4] = s.nbSpeech; qi[5] = s.nIn1; qi[6] = s.nIn2; qi[7] = s.nOut2;
    }
}

} // namespace sea
