/*
 * sea_tables.h -- constant tables of the MI355X noise-suppression engine, laid out lane-major
 * (one column per wavefront lane) so a kernel loads its per-lane constants with one coalesced
 * read each at start-up and then keeps them in VGPRs for the whole utterance.
 *
 * Built on the host by sea_tables.c (plain C, gcc -O2 -ffp-contract=off: the float/double
 * promotions of the reference's init code are reproduced exactly; citations there).
 */
#ifndef SEA_TABLES_H
#define SEA_TABLES_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
    SEA_LANES = 64,
    SEA_HOP = 80,          /* NS_FRAME_SHIFT      etsi/cpp/NoiseSup.h:52 */
    SEA_WIN = 200,         /* NS_FRAME_LENGTH     etsi/cpp/NoiseSup.h:54 */
    SEA_NFFT = 256,        /* NS_FFT_LENGTH       etsi/cpp/NoiseSup.h:43 */
    SEA_NSPEC = 65,        /* NS_SPEC_ORDER       etsi/cpp/NoiseSup.h:41 */
    SEA_NMEL = 25,         /* WF_MEL_ORDER        etsi/cpp/MelProcExports.h:18 */
    SEA_NTAP = 17,         /* NS_FILTER_LENGTH    etsi/cpp/NoiseSup.h:28 */
    SEA_MEL_TAPS = 10,     /* widest of the 25 Wiener mel bands (asserted at build) */
    SEA_FFT_LSTAGES = 6,   /* split-radix levels n2 = 8..256 that go through LDS */
    SEA_CC_NCHAN = 23,     /* CC_NUM_CHANNELS_WI8 etsi/cpp/CompCeps.c:37 */
    SEA_CC_TAPS = 22,      /* widest of the 23 cepstral mel triangles (asserted at build) */
    SEA_CC_TAPS2 = 24,     /* taps per lane of the tiled kernels' mel pass (aligned pairs, see melLaneBase) */
    SEA_CC_PWROW = 152,    /* words per power row of the tiled kernels: 129 bins, zeros behind */
    SEA_CC_NCEP = 14,      /* c1..c12, c0, logE   etsi/cpp/CompCeps.c:539-543 */
    SEA_GT_NCHAN = 64      /* NUMBER_CHANNEL      resyth_64sub_ori/cpp/HuWang.h:13 */
};

/* FFT work-item kinds for one split-radix level (etsi/cpp/rfft.c:99-178) */
enum { SEA_BF_NONE = 0, SEA_BF_PLAIN = 1, SEA_BF_PI4 = 2, SEA_BF_TWIDDLE = 3 };

typedef struct {
    /* rfft.c schedule, shared by NoiseSup and CompCeps.  Lane l owns the four bit-reversed
     * positions 4r..4r+3, r = bitrev6(l), which receive input elements l, l+128, l+64, l+192. */
    unsigned fftFlags[SEA_LANES];                 /* bit0: length-2 bf on (4r,4r+1); bit1: on
                                                     (4r+2,4r+3); bit2: n2=4 plain bf on group */
    unsigned fftItem[SEA_FFT_LSTAGES][SEA_LANES]; /* kind<<16 | b<<8 | a  (a=i1, b=i5) */
    float fftTw[SEA_FFT_LSTAGES][4][SEA_LANES];   /* cc1, ss1, cc3, ss3 of twiddle items */
    /* The same schedule packed into HALF a wavefront, for kernels that run two independent
     * transforms side by side (lanes 0..31 / 32..63): the plain and pi/4 butterflies of one block
     * are merged into one work item (kind SEA_BF_PAIR: a = plain base, b = pi/4 base), which
     * brings every level to at most 32 items. */
    unsigned fft2Item[SEA_FFT_LSTAGES][32];
    float fft2Tw[SEA_FFT_LSTAGES][4][32];
    /* LDS placement of the half-wave schedule.  Element i of a frame lives at word
     * sea_fft_swizzle(i) = i ^ kSwz[i >> 5] of its 256-word work area: with the natural layout the 32
     * lanes of a level hit 2..8 LDS banks (items sit 16, 32, ... words apart: 208 bank passes for
     * the 48 operand reads of a transform); with this XOR per 32-word block they hit 50.  All
     * addresses are table-driven, as BYTE offsets packed two per word (low | high << 16):
     *   fft2Addr[s][k][j]  operands 2k, 2k+1 of item j at level s, in the order
     *                      a, a+n4, a+2n4, a+3n4, b, b+n4, b+2n4, b+3n4
     *   fft2Head[k][l]     where lane l stores its bit-reversed head values 2k, 2k+1
     *   fft2Psd[k][l]      where lane l finds Re(2l), Re(2l+1) (k=0) and Im(2l+1) = x[255-2l],
     *                      Im(2l) = x[256-2l] (k=1; lane 0's second entry is unused)
     *   fft2Nyq            byte offset of x[128] */
    unsigned fft2Addr[SEA_FFT_LSTAGES][4][32];
    unsigned fftAddr[SEA_FFT_LSTAGES][4][SEA_LANES]; /* the same for the full-wave schedule (fftItem) */
    unsigned fft2Head[2][SEA_LANES];
    unsigned fft2Psd[2][SEA_LANES];
    unsigned fft2Nyq;
    unsigned fft2Pad[3];
    /* The dual transform's register-resident start: lane l (n0 = l & 31, transform l >> 5) owns the EIGHT
     * positions 8k..8k+7, k = bitrev5(n0), which receive input elements n0 + 32 * bitrev3(j), j = 0..7
     * (fft8Src[j] = 32 * bitrev3(j)), so that the length-2, n2 = 4 AND n2 = 8 butterflies all run on
     * registers: one LDS round trip fewer than the four-per-lane head + level n2 = 8.
     *   fft8Flags[l]    bits 0..3: length-2 butterfly on (8k+2p, 8k+2p+1); bits 4,5: n2 = 4 plain butterfly on
     *                   8k..8k+3 / 8k+4..8k+7; bit 6: n2 = 8 plain + pi/4 butterflies on the block
     *   fft8Addr[q][l]  swizzled byte offsets of positions 8k+2q, 8k+2q+1 (low | high << 16) */
    unsigned fft8Flags[SEA_LANES];
    unsigned fft8Addr[4][SEA_LANES];
    /* ... and the n2 = 16 level on registers too (round 4): the 16-block 16b..16b+15 is held by the lanes of 8-blocks k = 2b (n0 < 16)
     * and k = 2b + 1 (n0 >= 16 = the lane 16 further: v_permlane16_swap); the first takes the block's PAIR item (even positions:
     * plain butterfly on 16b + {0,4,8,12}, pi/4 on 16b + {2,6,10,14}), the second its one twiddled item (j = 1: 16b + {1,5,9,13},
     * 16b + {3,7,11,15}), each after receiving the other's four values of that parity.
     *   fft16Flags[l]   fft8Flags | bit 7: the block is in the n2 = 16 schedule
     *   fft16Addr[q][l] swizzled byte offsets of the lane's results 2q, 2q+1 in item order (the positions listed above) */
    unsigned fft16Flags[SEA_LANES];
    unsigned fft16Addr[4][SEA_LANES];
} sea_fft_tables;

/* word index of element i (0..255) in the swizzled work area */
static inline unsigned sea_fft_swizzle(unsigned i)
{
    static const unsigned char kSwz[8] = {0, 6, 29, 15, 18, 20, 9, 27};
    return i ^ kSwz[(i >> 5) & 7];
}

enum { SEA_BF_PAIR = 4 };

typedef struct {
    sea_fft_tables fft;
    float win[4][SEA_LANES];                      /* Hanning(200)[l+64k], 0 beyond 199 */
    float win8[8][SEA_LANES];                     /* Hanning(200)[(l & 31) + 32 * bitrev3(j)], 0 beyond 199: the dual head's layout */
    int melStart[SEA_LANES], melLen[SEA_LANES];   /* lanes 0..24 */
    float melW[SEA_MEL_TAPS][SEA_LANES];
    float idct[SEA_NMEL][SEA_LANES];              /* idct[f][t], lanes t = 0..8 */
    float irWin[SEA_LANES];                       /* lanes j = 0..8: Hanning(17)[8+j] */
    float eps;                                    /* NS_EPS = (float)exp(-10.0) */
    float pad[15];
} sea_ns_tables;

typedef struct {
    sea_fft_tables fft;
    float win[4][SEA_LANES];                      /* symmetric Hamming(200)[l+64k], 0 beyond */
    int melStart[SEA_LANES], melLen[SEA_LANES];   /* lanes 0..22 */
    float melW[SEA_CC_TAPS][SEA_LANES];
    float dct[SEA_CC_NCHAN][SEA_LANES];           /* dct[j][i-1], lanes i-1 = 0..11 */
    float floorFB, floorE;                        /* (float)exp(-10.0), (float)exp(-50.0) */
    float pad[14];
    /* the tiled kernels (cc_kernel.hip, cc_tile_*): two frames per wave through the dual transform */
    float win8[8][SEA_LANES];                     /* Hamming(200)[(l & 31) + 32 * bitrev3(j)], 0 beyond 199 */
    float dctT[SEA_CC_NCHAN][16];                 /* [j][c]: c = 0..11 the DCT rows of c1..c12, c = 12 all ones (c0 is
                                                     the plain sum of the 23 log energies: x * 1.0f == x), 13..15 zero */
    /* the mel pass of the tiled kernels, lane = one (frame of the pair, band): every lane reads SEA_CC_TAPS2 power bins as
     * SEA_CC_TAPS2 / 2 aligned pairs (ds_read_b64: 64 banks, two groups of 32 lanes) from an even bin at or below its band's
     * first one, zero weights in front and behind.  The 46 (frame, band) items are dealt to the lanes, and each picks its even
     * first bin, by a bipartite matching such that no two lanes of a group read the same pair of banks (cc_mel_lanes). */
    int melLaneBase[SEA_LANES];                   /* word offset from the pair's first power row (row h at SEA_CC_PWROW * h) */
    int melLaneFb[SEA_LANES];                     /* 24 h + band: where the lane's sum goes in fb[2 pr][..]; -1: idle lane */
    float melLaneW[SEA_CC_TAPS2][SEA_LANES];
} sea_cc_tables;

typedef struct {
    float gain[SEA_GT_NCHAN], f1[SEA_GT_NCHAN], f2[SEA_GT_NCHAN], midEar[SEA_GT_NCHAN];
    float cf[SEA_GT_NCHAN], bw[SEA_GT_NCHAN];
    double olaUp[160], olaDown[160];              /* 0.5(1+cos(n pi/160 + pi)), 0.5(1+cos(n pi/160)) */
} sea_gt_tables;

/* ---- rfft (x, n, m) for any size the reference's routine accepts (etsi/cpp/rfft.c:45-180) ----------------------------
 * The drop-in symbol `rfft` takes n = 2^q and any order m with 2^m <= n (the reference only ever passes (256, 8) and, in
 * the 16 k-native variant, (512, 8): order 8 on length 512, the last level never runs).  sea_rfft_schedule() unrolls the
 * routine's loop nest for one (n, m) into flat tables of 32-bit words for the one-workgroup kernel rfft_any_kernel:
 *   word 0 n, 1 m, 2 offset of rev[n] (input element e goes to place rev[e]: the digit-reverse counter, :57-79),
 *   3 offset / 4 count of the length-two butterflies' first elements (:82-96),
 *   for level k = 1 .. m-1 (n2 = 2^(k+1)):  5+3k offset / 6+3k count of the block starts i the is/id loops select
 *   (:107-130), 7+3k offset of the twiddles (cc1, ss1, cc3, ss3 as float bits for j = 0 .. n8-1; j = 0 unused; :133-138).
 * Returns a malloc'd array (caller frees) and its length in words, or NULL for a size the routine cannot take. */
enum { SEA_RFFT_MAXN = 16384, SEA_RFFT_HDR = 64 };
unsigned *sea_rfft_schedule(int n, int m, unsigned long *n_words);
int sea_rfft_schedule_host(float *x, int n, int m); /* the schedule walked on the CPU (tests) */

/* ---- the 16 k-native NoiseSup variant behind the reference's batch plug-in symbols (SURVEY 8(f) #4;
 * function/20141106_speech_enhancement/aurora_etsi/NoiseSup.h:36-53, NoiseSup.cpp:912-1407) ------------------------- */
enum {
    SEA16_HOP = 160,       /* NS_FRAME_SHIFT / NS_CUR_FRAME */
    SEA16_WIN = 480,       /* NS_FRAME_LENGTH */
    SEA16_AWIN = 80,       /* NS_ANALYSIS_WINDOW_8K: the window starts there in the 640-sample stage buffer */
    SEA16_BUF = 640,       /* NS_BUFFER_SIZE */
    SEA16_NFFT = 512,      /* NS_FFT_LENGTH, transformed with NS_FFT_ORDER 8: levels n2 = 4 .. 256 only */
    SEA16_NSPEC = 129,     /* NS_SPEC_ORDER */
    SEA16_NGAM = 25,       /* WF_MEL_ORDER: gammatone-shaped windows */
    SEA16_GLEN = 128,      /* every window covers gains 0..127 (MelProc.cpp:298) */
    SEA16_FFT_PASSES = 8,  /* pass 0: the length-2 butterflies; pass k: the L-shaped level n2 = 2^(k+1) */
    SEA16_FFT_SLOTS = 3    /* butterflies one lane runs per pass */
};
/* Slot s of pass p holds at most 64 butterflies of ONE kind, fixed per (p, s) so that the kernel's code per pass is
 * straight-line:  pass 0: three slots of length-2 butterflies (171);  pass 1 (n2 = 4): two slots of plain butterflies
 * (85; n4 = 1 has no pi/4 partner);  passes 2..7: slot 0 plain, slot 1 pi/4, slot 2 twiddled (none at n2 = 8).
 * Entry: valid << 31 | j << 16 | block start i. */
enum { SEA16_BF_LEN2 = 0, SEA16_BF_PLAIN = 1, SEA16_BF_PI4 = 2, SEA16_BF_TWIDDLE = 3 };

/* the pipelined kernel's transform wave (sea_tables.c::build_ns16k_pipe, ns16k_pipe_kernel.hip) */
enum { SEA16_PIPE_LEVELS = 5 }; /* n2 = 16 .. 256 through LDS; length-2, n2 = 4, n2 = 8 on registers */
#define SEA16_SWZ {0, 9, 22, 3, 10, 31, 5, 20, 18, 25, 16, 21, 11, 6, 12, 29} /* word i of the work area sits at i ^ SEA16_SWZ[i >> 5] */
typedef struct {
    unsigned head8Flags[SEA_LANES];                       /* bits 0..3 length-2 on (8l+2p, +1); 4, 5: n2 = 4 on 8l / 8l+4; 6: n2 = 8 on the block */
    unsigned head8Addr[4][SEA_LANES];                     /* swizzled byte offsets of places 8l+2q | 8l+2q+1 << 16 */
    unsigned kind[SEA16_PIPE_LEVELS][SEA_LANES];          /* SEA_BF_NONE / SEA_BF_PAIR / SEA_BF_TWIDDLE */
    unsigned addr[SEA16_PIPE_LEVELS][4][SEA_LANES];       /* the eight operands' byte offsets, two per word */
    float tw[SEA16_PIPE_LEVELS][4][SEA_LANES];
    unsigned psd[2][2][SEA_LANES];                        /* value l + 64 q: x[2b] | x[2b+1] << 16, x[512-2b] | x[511-2b] << 16 */
    unsigned nyq;                                         /* x[256] */
    unsigned pad[3];
    float win8[8][SEA_LANES];                             /* window weight of element src8[j][l] */
    unsigned short src8[8][SEA_LANES];                    /* bitrev6(l) + 64 bitrev3(j): the input element at place 8l + j */
} sea_ns16k_pipe_tables;

typedef struct {
    float sigWindow[SEA16_NFFT];                          /* Hanning(480), 0 beyond */
    unsigned short rev[SEA16_NFFT];                       /* where rfft.cpp's digit-reverse counter puts input element i */
    unsigned fftSlot[SEA16_FFT_PASSES][SEA16_FFT_SLOTS][SEA_LANES];
    float fftTw[SEA16_FFT_PASSES][32][4];                 /* cc1, ss1, cc3, ss3 of twiddle index j at that pass */
    float gammaT[SEA16_GLEN][SEA16_NGAM];                 /* [i][c]: window c's weight of gain i */
    float idctT[SEA16_NGAM][16];                          /* [f][t], t = 0..8: the nine taps DoFilterWindowing reads */
    float irWin[16];                                      /* [j] = Hanning(17)[8 + j], j = 0..8 */
    float eps;
    float pad[15];
    sea_ns16k_pipe_tables pipe;
} sea_ns16k_tables;
void sea_build_ns16k_tables(sea_ns16k_tables *t);
void sea_ns16k_fft_host(float *x512); /* the table-driven transform on the host (tests) */
void sea_ns16k_pipe_fft_host(float *x512); /* the same through the pipelined kernel's tables (tests) */
void sea_ns16k_plain_tables(float *sigWindow480, float *irWindow17, int *gammaStart25, float *gamma25x128, float *idct25x25);

void sea_build_ns_tables(sea_ns_tables *t);
void sea_build_cc_tables(sea_cc_tables *t);
void sea_build_gt_tables(sea_gt_tables *t);

/* plain tables for host-side checks (tests compare them with the oracle's) */
void sea_ns_plain_tables(float *sigWindow200, float *irWindow17, float *idct25x25, int *melStart25,
                         int *melLen25, float *melData /* 25*16 */);
void sea_cc_plain_tables(float *hamming100, float *dct12x23, int *melStart23, int *melLen23,
                         float *melData /* 23*32 */);

#ifdef __cplusplus
}
#endif
#endif
