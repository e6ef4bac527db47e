// gpsiq_kernels.hip — gfx950 (MI355X, wave64) kernels of libgpsiq.
//
// Replaces the per-sample loop of Mictronics/multi-sdr-gps-sim gps.c:2767-2836 and
// the pack of gps.c:2839-2846 with the closed-form integer NCO model of
// include/gpsiq.h.  One workgroup synthesises one tile of one 0.1 s block:
//   * the per-channel gain LUT  TC/TS[k] = (int)(table[k]*gain)  (gps.c:2781-2782
//     with dataBit*codeCA factored out) is built once per workgroup into LDS,
//     packed (I | Q<<16) as two int16 — sums are taken mod 2^16, which is exactly
//     what the reference's (short) store keeps (gps.c:2834-2835);
//   * the 1023-chip C/A codes are staged in LDS bit-packed, with a wrap-around tail;
//   * nothing is read from HBM per sample: the only traffic is the IQ write.
// No MFMA: there is no contraction in this path; it is integer VALU + LDS gather.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "gpsiq_internal.h"

namespace gpsiq {

constexpr int kMaxChan = GPSIQ_MAX_CHAN;
constexpr uint64_t kCodeFracMask = (UINT64_C(1) << GPSIQ_CODE_FRAC_BITS) - 1;

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int dev_sin512(const int16_t *qw, int k)
{
    k &= 511;
    int h = k & 255;
    int v = qw[h < 128 ? h : 255 - h];
    return k < 256 ? v : -v;
}

// Workgroup prologue shared by both kernels: descriptors, gain LUTs and C/A codes to LDS.
template <int NT>
__device__ __forceinline__ void stage_block(const gpsiq_qchan_t *__restrict__ q, int nchan,
                                            const DeviceTables *__restrict__ tab,
                                            gpsiq_qchan_t *qs, uint32_t (*lut)[512],
                                            uint32_t (*ext)[kPrnExtWords])
{
    const int tid = threadIdx.x;
    // 48-byte descriptors as 12 dwords each
    for (int i = tid; i < nchan * 12; i += NT)
        reinterpret_cast<uint32_t *>(qs)[i] = reinterpret_cast<const uint32_t *>(q)[i];
    __syncthreads();
    for (int e = tid; e < nchan * 512; e += NT) {
        const int c = e >> 9, k = e & 511;
        if (qs[c].prn == 0) continue;
        const double g = qs[c].gain;
        // (int)(int * double): exact int->double, one IEEE multiply, truncation toward 0
        const int ts = (int) ((double) dev_sin512(tab->quarter_wave, k) * g);
        const int tc = (int) ((double) dev_sin512(tab->quarter_wave, k + 128) * g);
        lut[c][k] = ((uint32_t) tc & 0xffffu) | ((uint32_t) ts << 16);
    }
    for (int e = tid; e < nchan * kPrnExtWords; e += NT) {
        const int c = e / kPrnExtWords, w = e % kPrnExtWords;
        if (qs[c].prn == 0) continue;
        ext[c][w] = tab->prn_ext[qs[c].prn - 1][w];
    }
    __syncthreads();
}

template <int FMT>
__device__ __forceinline__ void store_sample(uint8_t *__restrict__ blk_dst, uint32_t n, uint32_t iq)
{
    if (FMT == GPSIQ_SC16) {
        reinterpret_cast<uint32_t *>(blk_dst)[n] = iq;                      // gps.c:2842
    } else {
        // (signed char)(x >> 4) on each int16 half, arithmetic shift          gps.c:2845
        const int i16 = (int16_t) (iq & 0xffffu), q16 = (int16_t) (iq >> 16);
        reinterpret_cast<uint16_t *>(blk_dst)[n] =
            (uint16_t) (((uint32_t) (i16 >> 4) & 0xffu) | (((uint32_t) (q16 >> 4) & 0xffu) << 8));
    }
}

// ---------------------------------------------------------------------------
// Generic kernel: one sample per thread per step, every quantity of the closed form
// evaluated at full width for that sample.  Works for any rate the descriptor format
// allows; it is the fallback for sample rates too low for the row kernel, and an
// independent second implementation the tests cross-check the row kernel against.
constexpr int kGenericThreads = 256;

template <int FMT>
__global__ __launch_bounds__(kGenericThreads) void synth_generic(
    const gpsiq_qchan_t *__restrict__ desc, int nchan, int nsamp, uint8_t *__restrict__ dst,
    size_t block_stride, int block0, const DeviceTables *__restrict__ tab, int tiles_per_block,
    int tile_samples)
{
    __shared__ uint32_t lut[kMaxChan][512];
    __shared__ uint32_t ext[kMaxChan][kPrnExtWords];
    __shared__ gpsiq_qchan_t qs[kMaxChan];

    const int blk = blockIdx.x / tiles_per_block, tile = blockIdx.x % tiles_per_block;
    stage_block<kGenericThreads>(desc + (size_t) (block0 + blk) * nchan, nchan, tab, qs, lut, ext);
    uint8_t *blk_dst = dst + (size_t) blk * block_stride;

    const uint32_t n_begin = (uint32_t) tile * (uint32_t) tile_samples;
    uint32_t n_end = n_begin + (uint32_t) tile_samples;
    if (n_end > (uint32_t) nsamp) n_end = (uint32_t) nsamp;

    for (uint32_t n = n_begin + threadIdx.x; n < n_end; n += kGenericThreads) {
        int i_acc = 0, q_acc = 0;
        for (int c = 0; c < nchan; ++c) {
            const gpsiq_qchan_t &q = qs[c];
            if (q.prn == 0) continue;
            const uint64_t P = q.carr_phase + (uint64_t) q.carr_step * (uint64_t) n;
            const uint32_t idx = (uint32_t) (P >> (GPSIQ_CARR_FRAC_BITS - 9)) & 511u;
            const unsigned __int128 T = (unsigned __int128) q.code_frac +
                                        (unsigned __int128) q.code_step * (unsigned __int128) n;
            const uint64_t A = (uint64_t) q.chip0 + (uint64_t) (T >> GPSIQ_CODE_FRAC_BITS);
            const uint32_t chip = (uint32_t) (A % GPSIQ_CA_SEQ_LEN);
            const uint64_t period = A / GPSIQ_CA_SEQ_LEN;
            const uint32_t bit = (uint32_t) ((q.icode + period) / 20);
            const uint32_t neg = ((ext[c][chip >> 5] >> (chip & 31)) ^ (q.nav_bits >> (bit & 31))) & 1u;
            const uint32_t v = lut[c][idx];
            const int tc = (int16_t) (v & 0xffffu), ts = (int16_t) (v >> 16);
            i_acc += neg ? -tc : tc;
            q_acc += neg ? -ts : ts;
        }
        store_sample<FMT>(blk_dst, n, ((uint32_t) i_acc & 0xffffu) | ((uint32_t) q_acc << 16));
    }
}

// ---------------------------------------------------------------------------
// Row kernel.  A "row" is 64 consecutive samples, one per lane of a wave, so that
//   * IQ stores are one contiguous 128/256-byte run per wave instruction,
//   * the carrier LUT indices of a wave are (nearly) consecutive -> conflict-free
//     ds_read_b32 with broadcasts,
//   * all lanes of a row sit inside one 32-chip window of the C/A code, which is a
//     wave-uniform 32-bit word W (chip xor nav bit) prepared once per (channel,row):
//     W[A mod 32] is the sign of absolute chip A, so a lane needs only the low 5
//     bits of its own chip counter to find its sign — no mod-1023, no divisions.
// Each wave owns kRowsPerWave consecutive rows.  Per channel it keeps two 64-bit
// per-lane accumulators (carrier phase, code phase) and steps them by one row
// (64 samples) with a 64-bit add each; the per-lane start values are exact
// (64x32-bit products, wrapping mod 2^64 = mod 32 cycles / mod 256 chips).
// Code phase word: [chips mod 256 : 8][fraction : 56]; carrier word: [5 don't-care]
// [LUT index : 9][fraction : 50].
constexpr int kWaves = 8;
constexpr int kRowsPerWave = 32;
constexpr int kRowsThreads = kWaves * 64;
constexpr int kRowsTile = kWaves * kRowsPerWave * 64;   // 16384 samples
constexpr int kWinRowsPerLane = kRowsPerWave * kMaxChan / 64;   // rows one lane prepares

template <int FMT>
__global__ __launch_bounds__(kRowsThreads) void synth_rows(
    const gpsiq_qchan_t *__restrict__ desc, int nchan, int nsamp, uint8_t *__restrict__ dst,
    size_t block_stride, int block0, const DeviceTables *__restrict__ tab, int tiles_per_block)
{
    __shared__ uint32_t lut[kMaxChan][512];
    __shared__ uint32_t ext[kMaxChan][kPrnExtWords];
    __shared__ uint32_t win[kWaves][kMaxChan][kRowsPerWave];
    __shared__ gpsiq_qchan_t qs[kMaxChan];

    const int blk = blockIdx.x / tiles_per_block, tile = blockIdx.x % tiles_per_block;
    stage_block<kRowsThreads>(desc + (size_t) (block0 + blk) * nchan, nchan, tab, qs, lut, ext);
    uint8_t *blk_dst = dst + (size_t) blk * block_stride;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n_wave = (uint32_t) tile * kRowsTile + (uint32_t) wave * (kRowsPerWave * 64);

    // ---- windows: lane (c, g) prepares rows g*R .. g*R+R-1 of channel c -----------
    {
        const int c = lane & (kMaxChan - 1), g = lane / kMaxChan;
        if (c < nchan && qs[c].prn != 0) {
            const gpsiq_qchan_t &q = qs[c];
            const uint32_t n_row = n_wave + (uint32_t) (g * kWinRowsPerLane) * 64u;
            const unsigned __int128 T = (unsigned __int128) q.code_frac +
                                        (unsigned __int128) q.code_step * (unsigned __int128) n_row;
            uint64_t A = (uint64_t) q.chip0 + (uint64_t) (T >> GPSIQ_CODE_FRAC_BITS);
            uint64_t fr = (uint64_t) T & kCodeFracMask;
            uint32_t k = (uint32_t) (A % GPSIQ_CA_SEQ_LEN);          // chip inside the period
            const uint64_t ic = q.icode + A / GPSIQ_CA_SEQ_LEN;
            uint32_t bit = (uint32_t) (ic / 20), icur = (uint32_t) (ic % 20);
            uint32_t a5 = (uint32_t) A;                                // only A mod 32 is used
            const uint64_t row_step = q.code_step * 64u;
            const uint32_t d_int = (uint32_t) (row_step >> GPSIQ_CODE_FRAC_BITS);
            const uint64_t d_fr = row_step & kCodeFracMask;
            const uint32_t nav = q.nav_bits;
#pragma unroll
            for (int r = 0; r < kWinRowsPerLane; ++r) {
                // 32 chips starting at chip k of the (wrap-extended) code
                const uint32_t lo = ext[c][k >> 5], hi = ext[c][(k >> 5) + 1];
                uint32_t S = __builtin_amdgcn_alignbit(hi, lo, k & 31u);
                // chips at window positions >= 1023-k belong to the next code period
                const uint32_t to_wrap = GPSIQ_CA_SEQ_LEN - k;
                const uint32_t next_mask = to_wrap < 32u ? (0xffffffffu << to_wrap) : 0u;
                const uint32_t bit_next = icur == 19u ? bit + 1u : bit;
                const uint32_t d0 = 0u - ((nav >> (bit & 31u)) & 1u);
                const uint32_t d1 = 0u - ((nav >> (bit_next & 31u)) & 1u);
                S ^= (d0 & ~next_mask) ^ (d1 & next_mask);
                win[wave][c][g * kWinRowsPerLane + r] = __builtin_rotateleft32(S, a5 & 31u);
                // advance one row
                fr += d_fr;
                const uint32_t adv = d_int + (uint32_t) (fr >> GPSIQ_CODE_FRAC_BITS);
                fr &= kCodeFracMask;
                a5 += adv;
                k += adv;
                if (k >= GPSIQ_CA_SEQ_LEN) {
                    k -= GPSIQ_CA_SEQ_LEN;
                    if (++icur == 20u) { icur = 0u; ++bit; }
                }
            }
        }
    }
    __syncthreads();

    // ---- synthesis -------------------------------------------------------------
    s16x2 acc[kRowsPerWave];
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) acc[r] = (s16x2) (0);

    const uint32_t n0 = n_wave + (uint32_t) lane;
    for (int c = 0; c < nchan; ++c) {
        const gpsiq_qchan_t &q = qs[c];
        if (q.prn == 0) continue;
        uint64_t P = q.carr_phase + (uint64_t) q.carr_step * (uint64_t) n0;
        uint64_t Q = ((uint64_t) q.chip0 << GPSIQ_CODE_FRAC_BITS) + q.code_frac + q.code_step * (uint64_t) n0;
        const uint64_t dP = (uint64_t) q.carr_step * 64u;
        const uint64_t dQ = q.code_step * 64u;
        const unsigned char *lut_c = reinterpret_cast<const unsigned char *>(lut[c]);
        const uint32_t *win_c = win[wave][c];
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
            const uint32_t w = win_c[r];
            const uint32_t b = (uint32_t) (Q >> 56);
            const uint32_t m = (uint32_t) __builtin_amdgcn_sbfe((int) w, b, 1u);   // 0 or ~0
            const uint32_t sgn = m | 0x00010001u;                                 // (+1,+1) or (-1,-1)
            const uint32_t a = (uint32_t) (P >> 48) & 0x7fcu;                      // 4 * LUT index
            const uint32_t v = *reinterpret_cast<const uint32_t *>(lut_c + a);
            acc[r] = __builtin_bit_cast(s16x2, v) * __builtin_bit_cast(s16x2, sgn) + acc[r];
            P += dP;
            Q += dQ;
        }
    }

#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        const uint32_t n = n0 + (uint32_t) r * 64u;
        if (n < (uint32_t) nsamp)
            store_sample<FMT>(blk_dst, n, __builtin_bit_cast(uint32_t, acc[r]));
    }
}

// ---------------------------------------------------------------------------
// Row kernel, channel-inner order ("rowsx").  Same rows/windows/NCO words as synth_rows,
// but the loop nest is rows (outer) x channels (inner, fully unrolled over NCH slots):
//   * the per-lane NCO state of ALL channels stays in registers for the whole tile, so
//     the 64x32-bit start products are paid once per tile, not once per row group;
//   * the LUT base of every channel is a compile-time LDS offset (no address add);
//   * the NCH LUT gathers of a row are independent -> issued back to back, their LDS
//     latency overlaps instead of being exposed once per row;
//   * the NCH windows of a row are one contiguous 4*NCH-byte broadcast read.
// Descriptors arrive compacted (active channels first, see gpsiq_set_descriptors), the
// slots >= the block's active count are dummies: zero LUT, zero phase, zero step.
template <int FMT, int NCH>
__global__ __launch_bounds__(kRowsThreads) void synth_rowsx(
    const gpsiq_qchan_t *__restrict__ desc, int nchan, int nsamp, uint8_t *__restrict__ dst,
    size_t block_stride, int block0, const DeviceTables *__restrict__ tab, int tiles_per_block)
{
    __shared__ uint32_t lut[NCH][512];
    __shared__ uint32_t ext[NCH][kPrnExtWords];
    __shared__ uint32_t win[kWaves][kRowsPerWave][NCH];
    __shared__ gpsiq_qchan_t qs[NCH];

    const int tid = threadIdx.x;
    const int blk = blockIdx.x / tiles_per_block, tile = blockIdx.x % tiles_per_block;
    const gpsiq_qchan_t *q_blk = desc + (size_t) (block0 + blk) * nchan;
    const int nq = nchan < NCH ? nchan : NCH;
    for (int i = tid; i < NCH * 12; i += kRowsThreads)
        reinterpret_cast<uint32_t *>(qs)[i] = i < nq * 12 ? reinterpret_cast<const uint32_t *>(q_blk)[i] : 0u;
    __syncthreads();
    for (int e = tid; e < NCH * 512; e += kRowsThreads) {
        const int c = e >> 9, k = e & 511;
        uint32_t v = 0u;
        if (qs[c].prn != 0) {
            const double g = qs[c].gain;
            const int ts = (int) ((double) dev_sin512(tab->quarter_wave, k) * g);
            const int tc = (int) ((double) dev_sin512(tab->quarter_wave, k + 128) * g);
            v = ((uint32_t) tc & 0xffffu) | ((uint32_t) ts << 16);
        }
        lut[c][k] = v;
    }
    for (int e = tid; e < NCH * kPrnExtWords; e += kRowsThreads) {
        const int c = e / kPrnExtWords, w = e % kPrnExtWords;
        ext[c][w] = qs[c].prn ? tab->prn_ext[qs[c].prn - 1][w] : 0u;
    }
    __syncthreads();
    uint8_t *blk_dst = dst + (size_t) blk * block_stride;

    const int wave = tid >> 6, lane = tid & 63;
    const uint32_t n_wave = (uint32_t) tile * kRowsTile + (uint32_t) wave * (kRowsPerWave * 64);

    // ---- windows: lane (c, g) prepares a run of consecutive rows of channel c ------
    {
        constexpr int kGroups = 64 / NCH;                   // lanes per channel
        constexpr int kRun = kRowsPerWave / kGroups;        // rows per lane
        static_assert(kRowsPerWave % kGroups == 0, "rows per wave must split over the lane groups");
        const int c = lane % NCH, g = lane / NCH;
        if (g < kGroups) {
            const gpsiq_qchan_t &q = qs[c];
            const bool on = q.prn != 0;
            const uint32_t n_row = n_wave + (uint32_t) (g * kRun) * 64u;
            const unsigned __int128 T = (unsigned __int128) q.code_frac +
                                        (unsigned __int128) q.code_step * (unsigned __int128) n_row;
            uint64_t A = (uint64_t) q.chip0 + (uint64_t) (T >> GPSIQ_CODE_FRAC_BITS);
            uint64_t fr = (uint64_t) T & kCodeFracMask;
            uint32_t k = (uint32_t) (A % GPSIQ_CA_SEQ_LEN);
            const uint64_t ic = q.icode + A / GPSIQ_CA_SEQ_LEN;
            uint32_t bit = (uint32_t) (ic / 20), icur = (uint32_t) (ic % 20);
            uint32_t a5 = (uint32_t) A;
            const uint64_t row_step = q.code_step * 64u;
            const uint32_t d_int = (uint32_t) (row_step >> GPSIQ_CODE_FRAC_BITS);
            const uint64_t d_fr = row_step & kCodeFracMask;
            const uint32_t nav = q.nav_bits;
#pragma unroll 4
            for (int r = 0; r < kRun; ++r) {
                const uint32_t lo = ext[c][k >> 5], hi = ext[c][(k >> 5) + 1];
                uint32_t S = __builtin_amdgcn_alignbit(hi, lo, k & 31u);
                const uint32_t to_wrap = GPSIQ_CA_SEQ_LEN - k;
                const uint32_t next_mask = to_wrap < 32u ? (0xffffffffu << to_wrap) : 0u;
                const uint32_t bit_next = icur == 19u ? bit + 1u : bit;
                const uint32_t d0 = 0u - ((nav >> (bit & 31u)) & 1u);
                const uint32_t d1 = 0u - ((nav >> (bit_next & 31u)) & 1u);
                S ^= (d0 & ~next_mask) ^ (d1 & next_mask);
                win[wave][g * kRun + r][c] = on ? __builtin_rotateleft32(S, a5 & 31u) : 0u;
                fr += d_fr;
                const uint32_t adv = d_int + (uint32_t) (fr >> GPSIQ_CODE_FRAC_BITS);
                fr &= kCodeFracMask;
                a5 += adv;
                k += adv;
                if (k >= GPSIQ_CA_SEQ_LEN) {
                    k -= GPSIQ_CA_SEQ_LEN;
                    if (++icur == 20u) { icur = 0u; ++bit; }
                }
            }
        }
    }
    __syncthreads();

    // ---- per-lane NCO state of every channel --------------------------------------
    const uint32_t n0 = n_wave + (uint32_t) lane;
    uint64_t P[NCH], Q[NCH], dP[NCH], dQ[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        // descriptor fields through scalar loads (uniform address, read-only global):
        // the row steps must live in SGPRs, 4 per channel, or they cost 64 VGPRs
        const bool have = c < nchan;
        const uint64_t p0 = have ? q_blk[c].carr_phase : 0u, ps = have ? (uint64_t) q_blk[c].carr_step : 0u;
        const uint64_t f0 = have ? q_blk[c].code_frac : 0u, cs = have ? q_blk[c].code_step : 0u;
        const uint64_t c0 = have ? (uint64_t) q_blk[c].chip0 : 0u;
        P[c] = p0 + ps * (uint64_t) n0;
        Q[c] = (c0 << GPSIQ_CODE_FRAC_BITS) + f0 + cs * (uint64_t) n0;
        dP[c] = ps * 64u;
        dQ[c] = cs * 64u;
    }

    const unsigned char *lut_b = reinterpret_cast<const unsigned char *>(&lut[0][0]);
    for (int r = 0; r < kRowsPerWave; ++r) {
        const uint32_t n = n0 + (uint32_t) r * 64u;
        if (n_wave + (uint32_t) r * 64u >= (uint32_t) nsamp) break;     // wave-uniform
        const uint32_t *w_row = win[wave][r];
        s16x2 acc0 = (s16x2) (0), acc1 = (s16x2) (0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint32_t w = w_row[c];
            const uint32_t b = (uint32_t) (Q[c] >> 56);
            const uint32_t m = (uint32_t) __builtin_amdgcn_sbfe((int) w, b, 1u);
            const uint32_t sgn = m | 0x00010001u;
            const uint32_t a = (uint32_t) (P[c] >> 48) & 0x7fcu;
            const uint32_t v = *reinterpret_cast<const uint32_t *>(lut_b + c * 2048 + a);
            if (c & 1) acc1 = __builtin_bit_cast(s16x2, v) * __builtin_bit_cast(s16x2, sgn) + acc1;
            else       acc0 = __builtin_bit_cast(s16x2, v) * __builtin_bit_cast(s16x2, sgn) + acc0;
            P[c] += dP[c];
            Q[c] += dQ[c];
        }
        if (n < (uint32_t) nsamp)
            store_sample<FMT>(blk_dst, n, __builtin_bit_cast(uint32_t, acc0 + acc1));
    }
}

// ---------------------------------------------------------------------------
// Tile kernel ("tile"/"seg"): the row kernel in channel-inner order with the per-tile
// overhead trimmed, because the path is VALU-issue bound (profiles/r01_pmc_counters.txt:
// 9.7 VALU instructions per (channel,row) against 7 in the core):
//   * a wave owns `wave_rows` consecutive rows, worked in chunks of ROWS rows (the last chunk
//     may be partial), i.e. one contiguous run of samples:
//     the per-lane NCO words and the window-builder state simply continue from chunk to
//     chunk, so the start products, the mod-1023 / mod-20 set-up and the LUT build are
//     paid once per workgroup, not once per 64 rows ("seg": the host cuts every block into
//     equal runs of ~250 rows per wave; "tile": wave_rows = ROWS);
//   * LUT build: thread k owns LUT entry k of every channel, so sin/cos of k are formed
//     once and each entry costs two f64 multiplies and two truncations;
//   * window set-up in 32-bit chip arithmetic (a block never advances 2^32 chips);
//   * a chunk whose rows all lie inside the block runs a check-free row loop.
//   * H = 2 ("segh"): one window per HALF row (32 lanes), so a row may span up to 63 chips
//     and the kernel works down to one chip per sample (1.023 Msps); lanes 32..63 read the
//     second window of their row (two LDS addresses per wave read instead of one).
//   * FAST (int16 output: chosen by the host when no sum over the channels of a block can leave the
//     int16 range, i.e. sum of (int)(250*|gain|) <= 32767; int8 output: always): the LUT entry is the
//     single integer I + 65536*Q (int8: two 12-bit fields), the channel sum is a plain 32-bit add
//     (exact: it cannot overflow / the fields cannot collide), and the chip
//     sign is applied as half a carrier cycle: the carrier table is antisymmetric
//     (table[k+256] == -table[k], so is (int)(table*gain)), hence adding the sign bit to the top
//     index bit of the phase word selects the negated entry.  The shifted window's higher bits land
//     in the five spare bits above the index.  lshr, lshl_add, add replace bfe, or, pk_mad.
//   * BOTH ("segb", with FAST): the LUT holds both polarities, entry sign*512 + k = entry (k + 256*sign) mod 512, so the
//     chip sign is CONCATENATED above the index instead of added to it: with the phase word kept left-aligned (index in
//     bits 23..31 of its high half) one v_alignbit_b32 puts {sign, index} at bits 2..11 and a plain v_and_b32 with a
//     literal (2.5 issue cycles, where the SDWA form it replaces takes 4.3) makes the LDS address.  The table is 4 KB per
//     channel, so a workgroup has WAVES = 16 waves (64 KB of LUT + 64 KB of windows, one workgroup per CU = the same four
//     waves per SIMD).
template <int FMT, int NCH, int ROWS, int H, bool FAST, int WAVES = kWaves, bool BOTH = false>
__global__ __launch_bounds__(WAVES * 64, 4) void synth_tile(
    const gpsiq_qchan_t *__restrict__ desc, int nchan, int nsamp, uint8_t *__restrict__ dst,
    size_t block_stride, int block0, const DeviceTables *__restrict__ tab, int tiles_per_block,
    int wave_rows, int big_wgs, int big_blocks, int tiles_small)
{
    constexpr int kLutEntries = BOTH ? 1024 : 512;
    constexpr int kThreads = WAVES * 64;
    __shared__ uint32_t lut[NCH][kLutEntries];
    __shared__ uint32_t ext[NCH][kPrnExtWords];
    __shared__ uint32_t win[WAVES][ROWS * H][NCH];      // one window per (row or half row, channel)
    __shared__ gpsiq_qchan_t qs[NCH];
    static_assert(H == 1 || H == 2, "one window per row or per half row");
    static_assert(!BOTH || FAST, "the both-polarity table is a form of the plain-add core");
    static_assert(kLutEntries % kThreads == 0 || kThreads % kLutEntries == 0, "LUT build: whole passes");
    constexpr int kSpan = 64 / H;                        // samples per window

    const int tid = threadIdx.x;
    // workgroups [0, big_wgs) give every wave `wave_rows` consecutive rows (several chunks of
    // ROWS rows, the last one possibly partial) of blocks [0, big_blocks); the rest of the grid
    // covers the last blocks with one-chunk workgroups, so that what is still running when the
    // grid drains is short (workgroups are dispatched in id order)
    int blk, tile;
    if ((int) blockIdx.x < big_wgs) {
        blk = blockIdx.x / tiles_per_block; tile = blockIdx.x % tiles_per_block;
    } else {
        const int r = (int) blockIdx.x - big_wgs;
        blk = big_blocks + r / tiles_small; tile = r % tiles_small;
        wave_rows = ROWS;
    }
    const gpsiq_qchan_t *q_blk = desc + (size_t) (block0 + blk) * nchan;
    const int nq = nchan < NCH ? nchan : NCH;
    for (int i = tid; i < NCH * 12; i += kThreads)
        reinterpret_cast<uint32_t *>(qs)[i] = i < nq * 12 ? reinterpret_cast<const uint32_t *>(q_blk)[i] : 0u;
    __syncthreads();
    for (int e = tid; e < kLutEntries; e += kThreads) {
        // entry e of every channel (one pass: a thread per entry); unused slots have gain 0.0 -> entry 0.
        // BOTH: the upper half is the table half a cycle on, i.e. the negated entries
        const int k = BOTH ? (e + ((e >> 9) << 8)) & 511 : e;
        const double sk = (double) dev_sin512(tab->quarter_wave, k);
        const double ck = (double) dev_sin512(tab->quarter_wave, k + 128);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const double g = qs[c].gain;
            const int ts = (int) (sk * g), tc = (int) (ck * g);   // gps.c:2781-2782
            // int8 output keeps bits 4..11 of each 16-bit sum (gps.c:2845): with the entries
            // pre-shifted by 4 (still modulo 2^16) those bits are bytes 1 and 3 of the packed sum
            constexpr int kPre = FMT == GPSIQ_SC08 ? 4 : 0;
            if (FAST && FMT == GPSIQ_SC08)
                // the int8 output keeps bits 4..11 of I and Q only: 12-bit fields, I at bits 4..15 (its
                // carries spill into bits 16..19, 16 channels x 12 bits), Q at bits 20..31; the output
                // bytes are bytes 1 and 3 of the plain 32-bit sum, for any gain
                lut[c][e] = (((uint32_t) tc & 0xfffu) << 4) | ((uint32_t) ts << 20);
            else if (FAST)
                // one integer; |tc|, |ts| <= 32767 here.  Slot 0 also carries the +0x8000 that keeps
                // I + 32768 >= 0 in the sum (so a negative I never borrows from the Q half)
                lut[c][e] = (uint32_t) (tc + ts * 65536) + (c == 0 ? 0x8000u : 0u);
            else      lut[c][e] = (((uint32_t) tc << kPre) & 0xffffu) | ((uint32_t) ts << (16 + kPre));
        }
    }
    for (int e = tid; e < NCH * kPrnExtWords; e += kRowsThreads) {
        const int c = e / kPrnExtWords, w = e % kPrnExtWords;
        ext[c][w] = qs[c].prn ? tab->prn_ext[qs[c].prn - 1][w] : 0u;
    }
    __syncthreads();
    uint8_t *blk_dst = dst + (size_t) blk * block_stride;

    const int wave = tid >> 6, lane = tid & 63;
    const uint32_t wave_samples = (uint32_t) wave_rows * 64u;
    const uint32_t n_wave = ((uint32_t) tile * WAVES + (uint32_t) wave) * wave_samples;
    if (n_wave >= (uint32_t) nsamp) return;             // whole wave past the block end

    // ---- window builder: lane (c, g) prepares windows g, g+G, g+2G, ... of channel c ----
    constexpr int kPad = NCH <= 4 ? 4 : NCH <= 8 ? 8 : 16;   // lanes per window group
    constexpr int kGroups = 64 / kPad;
    constexpr int kRun = ROWS * H / kGroups;
    static_assert((ROWS * H) % kGroups == 0, "windows per chunk must split over the lane groups");
    const int c_raw = lane % kPad, wg = lane / kPad;
    const int wc = c_raw < NCH ? c_raw : 0;                  // surplus lanes shadow channel 0
    const bool w_store = c_raw < NCH;
    uint32_t w_k, w_nrev, w_nrot, w_dint;
    int32_t w_e1;
    uint64_t w_fr, w_dfr;
    {
        const gpsiq_qchan_t &q = qs[wc];
        const uint32_t n_row = n_wave + (uint32_t) wg * (uint32_t) kSpan;
        const unsigned __int128 T = (unsigned __int128) q.code_frac +
                                    (unsigned __int128) q.code_step * (unsigned __int128) n_row;
        const uint32_t A = (uint32_t) q.chip0 + (uint32_t) (uint64_t) (T >> GPSIQ_CODE_FRAC_BITS);
        w_fr = (uint64_t) T & kCodeFracMask;
        w_k = A % GPSIQ_CA_SEQ_LEN;                              // chip inside the period
        const uint32_t ic = q.icode + A / GPSIQ_CA_SEQ_LEN;
        // chips until the current nav bit ends, minus one (0..20459)
        w_e1 = (int32_t) ((20u - ic % 20u) * GPSIQ_CA_SEQ_LEN - w_k) - 1;
        // nav bits of this block, current bit in bit 31, the following ones below it
        w_nrev = __builtin_bitreverse32(q.nav_bits >> ((ic / 20u) & 31u));
        w_nrot = 0u - A;                                         // minus the rotation: only (-A) mod 32 matters
        // chips per builder step: kSpan*kGroups samples (<= 1024 samples at <= 0.5 chip, or
        // <= 512 samples at <= 1 chip: one period wrap at most)
        const unsigned __int128 step = (unsigned __int128) q.code_step * (unsigned) (kSpan * kGroups);
        w_dint = (uint32_t) (uint64_t) (step >> GPSIQ_CODE_FRAC_BITS);
        w_dfr = (uint64_t) step & kCodeFracMask;
    }

    // ---- per-lane NCO state of every channel (steps in SGPRs via scalar loads) -----
    const uint32_t n0 = n_wave + (uint32_t) lane;
    uint64_t P[NCH], Q[NCH], dP[NCH], dQ[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bool have = c < nchan;
        const uint64_t p0 = have ? q_blk[c].carr_phase : 0u, ps = have ? (uint64_t) q_blk[c].carr_step : 0u;
        const uint64_t f0 = have ? q_blk[c].code_frac : 0u, cs = have ? q_blk[c].code_step : 0u;
        const uint64_t c0 = have ? (uint64_t) q_blk[c].chip0 : 0u;
        // BOTH keeps the 59-bit phase left-aligned in the word (it then wraps by itself, index in the top nine bits)
        constexpr int kAlign = BOTH ? 64 - GPSIQ_CARR_FRAC_BITS : 0;
        P[c] = (p0 + ps * (uint64_t) n0) << kAlign;
        Q[c] = (c0 << GPSIQ_CODE_FRAC_BITS) + f0 + cs * (uint64_t) n0;
        dP[c] = (ps * 64u) << kAlign;
        dQ[c] = cs * 64u;
    }

    const unsigned char *lut_b = reinterpret_cast<const unsigned char *>(&lut[0][0]);
    const uint32_t *w_row = &win[wave][H == 2 ? lane >> 5 : 0][0];   // upper half wave: second window of the row
    uint32_t *w_dst = &win[wave][wg][wc];

    auto row_body = [&](int r, uint32_t n_chunk, bool check) {
        uint32_t iq;                                             // (I & 0xffff) | Q << 16, what the int16 store keeps
        if (FAST) {
            uint32_t sum = 0u;                                    // int16: slot 0's entries carry a +0x8000 bias
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t w = w_row[r * (H * NCH) + c];
                const uint32_t t = w >> ((uint32_t) (Q[c] >> 56) & 31u);        // bit 0 = chip ^ nav bit of this lane
                if (BOTH) {
                    // {strays, sign, index, 2 fraction bits}: sign and index make the word address in the 4 KB table
                    const uint32_t x = __builtin_amdgcn_alignbit(t, (uint32_t) (P[c] >> 32), 21u);
                    sum += *reinterpret_cast<const uint32_t *>(lut_b + c * 4096 + (x & 0xffcu));
                } else {
                    const uint32_t x = (t << 26) + (uint32_t) (P[c] >> 32);      // + half a cycle when that bit is set
                    const uint32_t a = (x >> 16) & 0x7fcu;
                    sum += *reinterpret_cast<const uint32_t *>(lut_b + c * 2048 + a);
                }
                P[c] += dP[c];
                Q[c] += dQ[c];
            }
            iq = FMT == GPSIQ_SC16 ? sum ^ 0x8000u : sum;         // take the bias off again: (I & 0xffff) | Q << 16
        } else {
            s16x2 acc0 = (s16x2) (0), acc1 = (s16x2) (0);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t w = w_row[r * (H * NCH) + c];
                const uint32_t b = (uint32_t) (Q[c] >> 56);
                const uint32_t m = (uint32_t) __builtin_amdgcn_sbfe((int) w, b, 1u);
                const uint32_t sgn = m | 0x00010001u;
                const uint32_t a = (uint32_t) (P[c] >> 48) & 0x7fcu;
                const uint32_t v = *reinterpret_cast<const uint32_t *>(lut_b + c * 2048 + a);
                if (c & 1) acc1 = __builtin_bit_cast(s16x2, v) * __builtin_bit_cast(s16x2, sgn) + acc1;
                else       acc0 = __builtin_bit_cast(s16x2, v) * __builtin_bit_cast(s16x2, sgn) + acc0;
                P[c] += dP[c];
                Q[c] += dQ[c];
            }
            iq = __builtin_bit_cast(uint32_t, acc0 + acc1);
        }
        const uint32_t n = n_chunk + (uint32_t) lane + (uint32_t) r * 64u;
        if (!check || n < (uint32_t) nsamp) {
            if (FMT == GPSIQ_SC16)
                *reinterpret_cast<uint32_t *>(blk_dst + n * 4u) = iq;                     // gps.c:2842
            else                                                                         // bytes 1 and 3, see the LUT build
                *reinterpret_cast<uint16_t *>(blk_dst + n * 2u) = (uint16_t) __builtin_amdgcn_perm(iq, iq, 0x0c0c0301u);
        }
    };

    for (int row0 = 0; row0 < wave_rows; row0 += ROWS) {
        const uint32_t n_chunk = n_wave + (uint32_t) row0 * 64u;
        // windows of this chunk (the builder state carries over from the previous chunk).
        // A nav-bit edge comes by once in 20 ms (about 800 rows) per channel: a group of four windows
        // during which no lane of the wave gets near one needs no attention to the edge counter and the
        // nav word (with 16 channels about two groups in three).
        constexpr int kGrp = 4;
        static_assert(kRun % kGrp == 0, "window groups");
        const uint32_t reach = (uint32_t) kGrp * (w_dint + 1u) + 32u;      // chips a lane can advance in a group + one window
#pragma unroll 1
        for (int i0 = 0; i0 < kRun; i0 += kGrp) {
            if (__builtin_amdgcn_ballot_w64((uint32_t) w_e1 <= reach) == 0) {
                const uint32_t nrot0 = w_nrot;
                const uint32_t d0 = (uint32_t) ((int32_t) w_nrev >> 31);
#pragma unroll
                for (int i = i0; i < i0 + kGrp; ++i) {
                    const uint32_t lo = ext[wc][w_k >> 5], hi = ext[wc][(w_k >> 5) + 1];
                    const uint32_t S = __builtin_amdgcn_alignbit(hi, lo, w_k) ^ d0;   // 32 chips from chip k (shift uses k & 31)
                    // rotate left by A mod 32 (alignbit rotates right by its low 5 bits); an unused slot
                    // needs no masking: its LUT entries are all zero
                    if (w_store) w_dst[i * (kGroups * NCH)] = __builtin_amdgcn_alignbit(S, S, w_nrot);
                    w_fr += w_dfr;
                    const uint32_t adv = w_dint + (uint32_t) (w_fr >> GPSIQ_CODE_FRAC_BITS);
                    w_fr &= kCodeFracMask;
                    w_nrot -= adv;
                    const uint32_t k2 = w_k + adv;                        // adv < 1023: one period wrap at most
                    w_k = k2 - GPSIQ_CA_SEQ_LEN < k2 ? k2 - GPSIQ_CA_SEQ_LEN : k2;
                }
                w_e1 -= (int32_t) (nrot0 - w_nrot);                       // chips advanced in this group
            } else {
#pragma unroll
                for (int i = i0; i < i0 + kGrp; ++i) {
                    const uint32_t lo = ext[wc][w_k >> 5], hi = ext[wc][(w_k >> 5) + 1];
                    uint32_t S = __builtin_amdgcn_alignbit(hi, lo, w_k);
                    S ^= (uint32_t) ((int32_t) w_nrev >> 31);
                    // the window holds the start of the next nav bit when fewer than 32 chips of the
                    // current one are left
                    const bool edge = (uint32_t) w_e1 < 31u;
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(edge) != 0, 0)) {
                        if (edge && ((w_nrev ^ (w_nrev << 1)) >> 31)) S ^= 0xfffffffeu << w_e1;
                    }
                    if (w_store) w_dst[i * (kGroups * NCH)] = __builtin_amdgcn_alignbit(S, S, w_nrot);
                    w_fr += w_dfr;
                    const uint32_t adv = w_dint + (uint32_t) (w_fr >> GPSIQ_CODE_FRAC_BITS);
                    w_fr &= kCodeFracMask;
                    w_nrot -= adv;
                    const uint32_t k2 = w_k + adv;
                    w_k = k2 - GPSIQ_CA_SEQ_LEN < k2 ? k2 - GPSIQ_CA_SEQ_LEN : k2;
                    const int32_t e = w_e1 - (int32_t) adv;               // adv < 20460: one bit edge at most
                    const int32_t m = e >> 31;
                    w_e1 = e + (m & (int32_t) (20 * GPSIQ_CA_SEQ_LEN));
                    w_nrev <<= (uint32_t) m & 1u;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        int rows = wave_rows - row0;                         // the wave's last chunk may be partial
        rows = rows < ROWS ? rows : ROWS;
        if (n_chunk + (uint32_t) rows * 64u <= (uint32_t) nsamp) {
            const int rows_s = __builtin_amdgcn_readfirstlane(rows);       // the trip count is wave-uniform: keep it scalar
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (int r = 0; r < rows_s; ++r) row_body(r, n_chunk, false);
        } else {                                             // the block ends inside this chunk
            const int in_block = (int) (((uint32_t) nsamp - n_chunk + 63u) >> 6);
            rows = rows < in_block ? rows : in_block;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (int r = 0; r < rows; ++r) row_body(r, n_chunk, true);
        }
        // the next chunk overwrites this wave's windows: all lanes must be done reading
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// ---------------------------------------------------------------------------
// Mask kernels ("segm"): an experiment for high sample rates, kept as a selectable, parity-tested variant; NOT the
// default, because it measured slower than seg (the row loop itself reaches its instruction
// count -- 0.78 ms against seg's 1.43 ms per 2 GiB at 25 Msps -- but delivering one 64-bit mask per (channel, row)
// costs more than it saves: +0.74 ms for streaming them through the scalar cache, +0.44 ms for the pre-pass).
// The idea: at >= 8 Msps a row of 64 samples holds only a few chip edges,
// so the chip-sign of a (channel, row) is a 64-bit lane mask that ONE thread can build bit-parallel by walking
// the edges with an exact DDA (sign_masks, a pre-pass: about ten instructions per edge, 64 (channel, row)s per
// wave instruction).  The row loop then needs no code NCO and no window per lane: the mask arrives through a
// scalar load as an EXEC mask, and the sign is applied as half a carrier cycle by one exec-masked
// v_xor_b32 a, 0x400, a on the LDS address (a plain two-operand VALU form).  Per (channel, row): carrier NCO add,
// SDWA and (address), masked xor, 1/2 add3 = 3.5 VALU instructions instead of 5.5, and nothing but the carrier
// LUT in LDS.  The arithmetic is the closed form of include/gpsiq.h, evaluated exactly:
//   sample pos of the chip edge e of a row: smallest pos with  f0 + pos*cs >= e*2^56  (f0: code fraction at the
//   row start), walked with  2^56 = q*cs + r:  the next edge is q or q+1 samples on, by the remainder.
constexpr int kMaskRowsPerThread = 16;      // consecutive rows one pre-pass thread walks (one division per 1024 samples)

__global__ __launch_bounds__(256) void sign_masks(
    const gpsiq_qchan_t *__restrict__ desc, int nchan, int nsamp, int block0, int nblocks,
    const DeviceTables *__restrict__ tab, uint64_t *__restrict__ masks, int rows_total, int rowgroups)
{
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int) (t & 15u);
    const unsigned g = t >> 4;
    const int rg = (int) (g % (unsigned) rowgroups), blk = (int) (g / (unsigned) rowgroups);
    if (blk >= nblocks) return;
    const int row0 = rg * kMaskRowsPerThread;
    uint64_t *out = masks + ((size_t) blk * rows_total + row0) * 16 + c;
    int nrows = rows_total - row0;
    if (nrows > kMaskRowsPerThread) nrows = kMaskRowsPerThread;
    if (c >= nchan || desc[(size_t) (block0 + blk) * nchan + c].prn == 0) {     // unused slot: its LUT is all zero
        for (int r = 0; r < nrows; ++r) out[(size_t) r * 16] = 0u;
        return;
    }
    const gpsiq_qchan_t q = desc[(size_t) (block0 + blk) * nchan + c];
    const uint32_t *prn = tab->prn_ext[q.prn - 1];
    const uint64_t cs = q.code_step;
    const unsigned __int128 T = (unsigned __int128) q.code_frac + (unsigned __int128) cs * (unsigned) (row0 * 64);
    const uint32_t A0 = (uint32_t) q.chip0 + (uint32_t) (uint64_t) (T >> GPSIQ_CODE_FRAC_BITS);
    const uint64_t f0 = (uint64_t) T & kCodeFracMask;
    uint32_t k = A0 % GPSIQ_CA_SEQ_LEN;
    const uint32_t ic = q.icode + A0 / GPSIQ_CA_SEQ_LEN;
    uint32_t bit = ic / 20u, icur = ic % 20u;
    const uint64_t one = UINT64_C(1) << GPSIQ_CODE_FRAC_BITS;
    const uint64_t q56 = one / cs, r56 = one % cs;
    // first edge after the group's first sample: smallest pos with f0 + pos*cs >= 2^56
    const uint64_t d = one - f0;
    uint64_t pos = (d + cs - 1) / cs;
    uint64_t over = pos * cs - d;                         // how far past the edge that sample is, < cs
    uint32_t sgn = ((prn[k >> 5] >> (k & 31u)) ^ (q.nav_bits >> (bit & 31u))) & 1u;
    for (int r = 0; r < nrows; ++r) {
        uint64_t m = sgn ? ~UINT64_C(0) : UINT64_C(0);
        while (pos < 64u) {
            if (++k == GPSIQ_CA_SEQ_LEN) {                 // gps.c:2791-2811: next code period, maybe next data bit
                k = 0;
                if (++icur == 20u) { icur = 0; ++bit; }
            }
            const uint32_t ns = ((prn[k >> 5] >> (k & 31u)) ^ (q.nav_bits >> (bit & 31u))) & 1u;
            if (ns != sgn) { m ^= ~UINT64_C(0) << pos; sgn = ns; }
            if (r56 > over) { pos += q56 + 1; over += cs - r56; }
            else            { pos += q56;     over -= r56; }
        }
        out[(size_t) r * 16] = m;
        pos -= 64u;
    }
}

template <int FMT, int NCH>
__global__ __launch_bounds__(kRowsThreads) void synth_mask(
    const gpsiq_qchan_t *__restrict__ desc, int nchan, int nsamp, uint8_t *__restrict__ dst,
    size_t block_stride, int block0, const DeviceTables *__restrict__ tab, const uint64_t *__restrict__ masks,
    int rows_total, int tiles_per_block, int wave_rows)
{
    __shared__ uint32_t lut[NCH][512];
    __shared__ gpsiq_qchan_t qs[NCH];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x / tiles_per_block, tile = blockIdx.x % tiles_per_block;
    const gpsiq_qchan_t *q_blk = desc + (size_t) (block0 + blk) * nchan;
    const int nq = nchan < NCH ? nchan : NCH;
    for (int i = tid; i < NCH * 12; i += kRowsThreads)
        reinterpret_cast<uint32_t *>(qs)[i] = i < nq * 12 ? reinterpret_cast<const uint32_t *>(q_blk)[i] : 0u;
    __syncthreads();
    {
        const double sk = (double) dev_sin512(tab->quarter_wave, tid);
        const double ck = (double) dev_sin512(tab->quarter_wave, tid + 128);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const double g = qs[c].gain;
            const int ts = (int) (sk * g), tc = (int) (ck * g);   // gps.c:2781-2782
            // the plain-add formats of synth_tile<FAST>: int8 two 12-bit fields, int16 one integer with slot 0's bias
            if (FMT == GPSIQ_SC08) lut[c][tid] = (((uint32_t) tc & 0xfffu) << 4) | ((uint32_t) ts << 20);
            else                   lut[c][tid] = (uint32_t) (tc + ts * 65536) + (c == 0 ? 0x8000u : 0u);
        }
    }
    __syncthreads();
    uint8_t *blk_dst = dst + (size_t) blk * block_stride;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int row_first = (tile * kWaves + wave) * wave_rows;
    if (row_first >= rows_total) return;
    int rows = rows_total - row_first;
    rows = rows < wave_rows ? rows : wave_rows;
    const uint32_t n0 = (uint32_t) row_first * 64u + (uint32_t) lane;
    uint64_t P[NCH], dP[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bool have = c < nchan;
        const uint64_t p0 = have ? q_blk[c].carr_phase : 0u, ps = have ? (uint64_t) q_blk[c].carr_step : 0u;
        P[c] = p0 + ps * (uint64_t) n0;
        dP[c] = ps * 64u;
    }
    const unsigned char *lut_b = reinterpret_cast<const unsigned char *>(&lut[0][0]);
    // the 16 masks of a row are one 128-byte line at a wave-uniform address: two s_load_dwordx16 per row
    typedef uint64_t u64x8 __attribute__((ext_vector_type(8)));
    const u64x8 *m_row = reinterpret_cast<const u64x8 *>(masks + ((size_t) blk * rows_total + row_first) * 16);
    const int rows_s = __builtin_amdgcn_readfirstlane(rows);
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int r = 0; r < rows_s; ++r) {
        const u64x8 m_lo = m_row[2 * r];
        u64x8 m_hi = m_lo;
        if (NCH > 8) m_hi = m_row[2 * r + 1];
        uint32_t sum = 0u;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint64_t m = c < 8 ? m_lo[c] : m_hi[c - 8];
            uint32_t a = (uint32_t) (P[c] >> 48) & 0x7fcu;                      // 4 * LUT index
            uint64_t saved;
            // lanes whose chip sign is -1 read the entry half a cycle on: the table is antisymmetric
            asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tv_xor_b32 %[a], 0x400, %[a]\n\ts_mov_b64 exec, %[sv]"
                         : [a] "+v"(a), [sv] "=&s"(saved) : [m] "s"(m));
            sum += *reinterpret_cast<const uint32_t *>(lut_b + c * 2048 + a);
            P[c] += dP[c];
        }
        const uint32_t iq = FMT == GPSIQ_SC16 ? sum ^ 0x8000u : sum;
        const uint32_t n = n0 + (uint32_t) r * 64u;
        if (n < (uint32_t) nsamp) {
            if (FMT == GPSIQ_SC16) *reinterpret_cast<uint32_t *>(blk_dst + n * 4u) = iq;
            else *reinterpret_cast<uint16_t *>(blk_dst + n * 2u) = (uint16_t) __builtin_amdgcn_perm(iq, iq, 0x0c0c0301u);
        }
    }
}

size_t variant_scratch_bytes(int variant, int nsamp, int nblocks)
{
    if (variant != kSegMask || nsamp <= 0 || nblocks <= 0) return 0;
    return (size_t) nblocks * (size_t) ((nsamp + 63) / 64) * 16u * sizeof(uint64_t);
}

// ---------------------------------------------------------------------------
// GPSIQ_NCO_REFERENCE fix-up (csrc/gpsiq_exact.cpp): the few samples per 10^7 where the reference's
// double accumulators pick another LUT entry or sign than the closed form are recomputed whole --
// every channel from the closed form, the patched channels from the patch -- and stored over what
// the synthesis kernel wrote (same stream, after it).  Sixteen lanes per patch, one channel each (the
// closed form of one channel is a chain of 128-bit products and a division: sixteen of them one
// after the other in one thread were 13-18 us of a 200 us piece), summed with cross-lane
// shuffles inside each group of sixteen; the first patch of a (block, sample) group does the sample, lane 0 stores it.
template <int FMT>
__global__ __launch_bounds__(64) void apply_patches(
    const gpsiq_qchan_t *__restrict__ desc, int nchan, int nsamp, uint8_t *__restrict__ dst, size_t block_stride,
    int block0, int nblocks, const DeviceTables *__restrict__ tab, const gpsiq_patch_t *__restrict__ pt, int npatch)
{
    const int i = (int) (blockIdx.x * 4u + (threadIdx.x >> 4));      // four patches per wave
    const int c = (int) (threadIdx.x & 15u);                          // this lane's channel slot (GPSIQ_MAX_CHAN = 16)
    const bool have = i < npatch;
    const gpsiq_patch_t p = pt[have ? i : npatch - 1];
    bool lead = have && !(i > 0 && pt[i - 1].block == p.block && pt[i - 1].sample == p.sample);
    lead = lead && p.block >= (uint32_t) block0 && p.block < (uint32_t) (block0 + nblocks) && p.sample < (uint32_t) nsamp;
    int i_acc = 0, q_acc = 0;
    if (lead && c < nchan) {
        const gpsiq_qchan_t qc = desc[(size_t) p.block * nchan + c];
        if (qc.prn != 0) {
            const uint64_t n = p.sample;
            const uint64_t P = qc.carr_phase + (uint64_t) qc.carr_step * n;
            uint32_t idx = (uint32_t) (P >> (GPSIQ_CARR_FRAC_BITS - 9)) & 511u;
            const unsigned __int128 T = (unsigned __int128) qc.code_frac + (unsigned __int128) qc.code_step * (unsigned __int128) n;
            const uint64_t A = (uint64_t) qc.chip0 + (uint64_t) (T >> GPSIQ_CODE_FRAC_BITS);
            const uint32_t chip = (uint32_t) (A % GPSIQ_CA_SEQ_LEN);
            const uint32_t bit = (uint32_t) ((qc.icode + A / GPSIQ_CA_SEQ_LEN) / 20);
            uint32_t neg = ((tab->prn_ext[qc.prn - 1][chip >> 5] >> (chip & 31)) ^ (qc.nav_bits >> (bit & 31))) & 1u;
            for (int j = i; j < npatch && pt[j].block == p.block && pt[j].sample == p.sample; ++j)
                if (pt[j].slot == c) { idx = pt[j].lut & 511u; neg = pt[j].neg & 1u; }
            const int ts = (int) ((double) dev_sin512(tab->quarter_wave, (int) idx) * qc.gain);        // gps.c:2782
            const int tc = (int) ((double) dev_sin512(tab->quarter_wave, (int) idx + 128) * qc.gain);  // gps.c:2781
            i_acc = neg ? -tc : tc;
            q_acc = neg ? -ts : ts;
        }
    }
    // sum over the sixteen lanes of the patch (all 64 lanes take part: no divergence around the cross-lane moves)
    for (int off = 8; off >= 1; off >>= 1) {
        i_acc += __shfl_xor(i_acc, off, 16);
        q_acc += __shfl_xor(q_acc, off, 16);
    }
    if (lead && c == 0)
        store_sample<FMT>(dst + (size_t) (p.block - (uint32_t) block0) * block_stride, p.sample,
                          ((uint32_t) i_acc & 0xffffu) | ((uint32_t) q_acc << 16));
}

hipError_t launch_patches(const gpsiq_qchan_t *desc, int nchan, int nsamp, int sample_size, void *dst, size_t block_stride,
                          int block0, int nblocks, const DeviceTables *tab, const gpsiq_patch_t *patches, int npatch,
                          hipStream_t stream)
{
    if (npatch <= 0 || nblocks <= 0 || nsamp <= 0) return hipSuccess;
    dim3 grid((unsigned) ((npatch + 3) / 4)), block(64);
    uint8_t *d = static_cast<uint8_t *>(dst);
    if (sample_size == GPSIQ_SC16)
        hipLaunchKernelGGL(apply_patches<GPSIQ_SC16>, grid, block, 0, stream, desc, nchan, nsamp, d, block_stride, block0, nblocks, tab, patches, npatch);
    else
        hipLaunchKernelGGL(apply_patches<GPSIQ_SC08>, grid, block, 0, stream, desc, nchan, nsamp, d, block_stride, block0, nblocks, tab, patches, npatch);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Grid-shape policy of the seg variants; the defaults can be overridden for experiments with
// GPSIQ_SEG_TAIL_WGS / GPSIQ_SEG_MAX_WAVE_ROWS / GPSIQ_SEG_SETUP_ROWS / GPSIQ_SEG_DRAIN (read once).
struct SegPolicy {
    int    tail_wgs;        // one-chunk workgroups at the end of the grid
    int    max_wave_rows;   // longest run of rows a wave may own
    double setup_rows;      // per-workgroup set-up, in row-times (measured: tile vs seg = 4 %)
    double drain_rounds;    // time lost while the grid drains, in workgroup durations
    double resident_wgs;    // 256 CUs x 2 workgroups (67 KB LDS each)
    bool   allow_fast;      // GPSIQ_NO_FAST=1 forces the packed-multiply kernels (A/B experiments, tests)
};
static const SegPolicy &seg_policy()
{
    static const SegPolicy pol = [] {
        SegPolicy p = {512, 512, 3.5, 0.3, 512.0, true};
        if (const char *e = std::getenv("GPSIQ_SEG_TAIL_WGS")) p.tail_wgs = std::atoi(e);
        if (const char *e = std::getenv("GPSIQ_SEG_MAX_WAVE_ROWS")) p.max_wave_rows = std::atoi(e);
        if (const char *e = std::getenv("GPSIQ_SEG_SETUP_ROWS")) p.setup_rows = std::atof(e);
        if (const char *e = std::getenv("GPSIQ_SEG_DRAIN")) p.drain_rounds = std::atof(e);
        if (const char *e = std::getenv("GPSIQ_NO_FAST")) p.allow_fast = std::atoi(e) == 0;
        return p;
    }();
    return pol;
}

hipError_t launch_variant(int variant, const gpsiq_qchan_t *desc, int nchan, int nsamp, int sample_size,
                          void *dst, size_t block_stride, int block0, int nblocks,
                          const DeviceTables *tab, hipStream_t stream, int max_active, long max_amplitude, void *scratch)
{
    if (nblocks <= 0 || nsamp <= 0) return hipSuccess;
    uint8_t *d = static_cast<uint8_t *>(dst);
    // the mask kernel only has the plain-add LUT formats: int16 sums that may leave the int16 range go to seg's packed core
    if (variant == kSegMask && ((sample_size == GPSIQ_SC16 && max_amplitude > 32767) || !scratch)) variant = kSeg;
    if (variant == kSegMask) {
        const int rows_total = (nsamp + 63) / 64;
        const int rowgroups = (rows_total + kMaskRowsPerThread - 1) / kMaskRowsPerThread;
        uint64_t *masks = static_cast<uint64_t *>(scratch);
        const size_t threads = (size_t) nblocks * rowgroups * 16;
        hipLaunchKernelGGL(sign_masks, dim3((unsigned) ((threads + 255) / 256)), dim3(256), 0, stream, desc, nchan, nsamp, block0, nblocks,
                           tab, masks, rows_total, rowgroups);
        // every wave the same number of rows; workgroups of ~256 rows per wave amortise the LUT build
        int wave_rows = 256, tiles = (rows_total + kWaves * wave_rows - 1) / (kWaves * wave_rows);
        wave_rows = (rows_total + kWaves * tiles - 1) / (kWaves * tiles);
        dim3 grid((unsigned) (tiles * nblocks)), block(kRowsThreads);
        const int slots = max_active <= 4 ? 4 : max_active <= 8 ? 8 : max_active <= 12 ? 12 : 16;
#define GPSIQ_LAUNCH_M(F, N) hipLaunchKernelGGL((synth_mask<F, N>), grid, block, 0, stream, desc, nchan, nsamp, d, block_stride, block0, tab, masks, rows_total, tiles, wave_rows)
        if (sample_size == GPSIQ_SC16) {
            if (slots == 4) GPSIQ_LAUNCH_M(GPSIQ_SC16, 4); else if (slots == 8) GPSIQ_LAUNCH_M(GPSIQ_SC16, 8);
            else if (slots == 12) GPSIQ_LAUNCH_M(GPSIQ_SC16, 12); else GPSIQ_LAUNCH_M(GPSIQ_SC16, 16);
        } else {
            if (slots == 4) GPSIQ_LAUNCH_M(GPSIQ_SC08, 4); else if (slots == 8) GPSIQ_LAUNCH_M(GPSIQ_SC08, 8);
            else if (slots == 12) GPSIQ_LAUNCH_M(GPSIQ_SC08, 12); else GPSIQ_LAUNCH_M(GPSIQ_SC08, 16);
        }
#undef GPSIQ_LAUNCH_M
        return hipGetLastError();
    }
    // the both-polarity table is a form of the plain-add core: sums that may leave the int16 range go to seg's packed core
    if (variant == kSegBoth && !((sample_size == GPSIQ_SC08 || max_amplitude <= 32767) && seg_policy().allow_fast)) variant = kSeg;
    if (variant == kSegBoth) {
        // 4 KB of LUT per channel: with 16 channels two 8-wave workgroups still fit a CU (the same four waves per SIMD as
        // seg) when a chunk is 16 rows, i.e. 8 KB of windows per workgroup (64 + 8 + 3 KB, twice = 150 of 160 KB); a single
        // 16-wave workgroup per CU with 64-row chunks measured 10 % SLOWER than seg although its waves ran 11 % faster
        // (SQ_WAVE_CYCLES): with one workgroup per CU nothing fills the CU while that workgroup starts up or drains.
        const int slots = max_active <= 4 ? 4 : max_active <= 8 ? 8 : max_active <= 12 ? 12 : 16;
        const int rows = slots == 4 ? 64 : slots == 8 ? 32 : 16;     // rows per chunk: the lane groups of the window builder need 4 windows each
        const SegPolicy &pol = seg_policy();
        const int rows_total = (nsamp + 63) / 64;
        const int tiles1 = (rows_total + kWaves * rows - 1) / (kWaves * rows);
        int wave_rows = rows, tiles = tiles1, tail_blocks = 0;
        double best = 0.0;
        for (int nwg = 1; nwg <= tiles1; ++nwg) {
            const int wr = (rows_total + kWaves * nwg - 1) / (kWaves * nwg);
            if (wr > pol.max_wave_rows) continue;
            if (wr < rows && nwg < tiles1) break;
            const double fill = (double) rows_total / ((double) kWaves * nwg * wr);
            const double amort = (double) wr / ((double) wr + pol.setup_rows);
            const double rounds = (double) nblocks * nwg / pol.resident_wgs;
            const double score = fill * amort * rounds / (rounds + pol.drain_rounds);
            if (score > best) { best = score; wave_rows = wr > rows ? wr : rows; tiles = nwg; }
        }
        if (wave_rows > rows) {
            tail_blocks = (pol.tail_wgs + tiles1 - 1) / tiles1;
            if (tail_blocks > nblocks / 2) tail_blocks = nblocks / 2;
        }
        const int big_blocks = nblocks - tail_blocks;
        const int big_wgs = tiles * big_blocks;
        dim3 grid((unsigned) (big_wgs + tiles1 * tail_blocks)), block(kRowsThreads);
#define GPSIQ_LAUNCH_B(F, N, R) hipLaunchKernelGGL((synth_tile<F, N, R, 1, true, kWaves, true>), grid, block, 0, stream, desc, nchan, nsamp, d, block_stride, block0, tab, tiles, wave_rows, big_wgs, big_blocks, tiles1)
        if (sample_size == GPSIQ_SC16) {
            if (slots == 4) GPSIQ_LAUNCH_B(GPSIQ_SC16, 4, 64); else if (slots == 8) GPSIQ_LAUNCH_B(GPSIQ_SC16, 8, 32);
            else if (slots == 12) GPSIQ_LAUNCH_B(GPSIQ_SC16, 12, 16); else GPSIQ_LAUNCH_B(GPSIQ_SC16, 16, 16);
        } else {
            if (slots == 4) GPSIQ_LAUNCH_B(GPSIQ_SC08, 4, 64); else if (slots == 8) GPSIQ_LAUNCH_B(GPSIQ_SC08, 8, 32);
            else if (slots == 12) GPSIQ_LAUNCH_B(GPSIQ_SC08, 12, 16); else GPSIQ_LAUNCH_B(GPSIQ_SC08, 16, 16);
        }
#undef GPSIQ_LAUNCH_B
        return hipGetLastError();
    }
    if (variant == kTile || variant == kSeg || variant == kSegHalf) {
        const bool half = variant == kSegHalf;
        const int rows = half ? 32 : 64;                  // rows per chunk (the window array holds 64 windows per wave)
        const int rows_total = (nsamp + 63) / 64;
        // seg: many rows per wave amortise the per-workgroup set-up (LUT build, start products),
        // but long workgroups make the drain of the grid expensive.  Every block is cut into
        // nwg workgroups whose 8 waves all get the same number of rows, so no wave idles while
        // its workgroup holds a CU slot, whatever the block length.  nwg maximises
        //   (rows used / rows scheduled) x (rows per wave / (rows per wave + set-up)) x (rounds / (rounds + drain)),
        // a model fitted to the measured variant sweeps; the last blocks are covered by
        // one-chunk workgroups so that the drain is short.
        const int tiles1 = (rows_total + kWaves * rows - 1) / (kWaves * rows);
        int wave_rows = rows, tiles = tiles1, tail_blocks = 0;
        if (variant != kTile) {
            const SegPolicy &pol = seg_policy();
            double best = 0.0;
            for (int nwg = 1; nwg <= tiles1; ++nwg) {
                const int wr = (rows_total + kWaves * nwg - 1) / (kWaves * nwg);
                if (wr > pol.max_wave_rows) continue;
                if (wr < rows && nwg < tiles1) break;
                const double fill = (double) rows_total / ((double) kWaves * nwg * wr);
                const double amort = (double) wr / ((double) wr + pol.setup_rows);
                const double rounds = (double) nblocks * nwg / pol.resident_wgs;
                const double score = fill * amort * rounds / (rounds + pol.drain_rounds);
                if (score > best) { best = score; wave_rows = wr > rows ? wr : rows; tiles = nwg; }
            }
            if (wave_rows > rows) {
                tail_blocks = (pol.tail_wgs + tiles1 - 1) / tiles1;
                if (tail_blocks > nblocks / 2) tail_blocks = nblocks / 2;
            }
        }
        const int big_blocks = nblocks - tail_blocks;
        const int big_wgs = tiles * big_blocks;
        dim3 grid((unsigned) (big_wgs + tiles1 * tail_blocks)), block(kRowsThreads);
        // no channel sum of any resident block can leave the int16 range: plain-add kernel
        // (the int8 kernels keep 12-bit fields and are exact for any gain)
        const bool fast = (sample_size == GPSIQ_SC08 || max_amplitude <= 32767) && seg_policy().allow_fast;
#define GPSIQ_LAUNCH_T4(F, N, R, HH, FA) hipLaunchKernelGGL((synth_tile<F, N, R, HH, FA>), grid, block, 0, stream, desc, nchan, nsamp, d, block_stride, block0, tab, tiles, wave_rows, big_wgs, big_blocks, tiles1)
#define GPSIQ_LAUNCH_T(F, N) do { if (half) { if (fast) GPSIQ_LAUNCH_T4(F, N, 32, 2, true); else GPSIQ_LAUNCH_T4(F, N, 32, 2, false); } \
                                  else      { if (fast) GPSIQ_LAUNCH_T4(F, N, 64, 1, true); else GPSIQ_LAUNCH_T4(F, N, 64, 1, false); } } while (0)
        const int slots = max_active <= 4 ? 4 : max_active <= 8 ? 8 : max_active <= 12 ? 12 : 16;
        if (sample_size == GPSIQ_SC16) {
            if (slots == 4) GPSIQ_LAUNCH_T(GPSIQ_SC16, 4); else if (slots == 8) GPSIQ_LAUNCH_T(GPSIQ_SC16, 8);
            else if (slots == 12) GPSIQ_LAUNCH_T(GPSIQ_SC16, 12); else GPSIQ_LAUNCH_T(GPSIQ_SC16, 16);
        } else {
            if (slots == 4) GPSIQ_LAUNCH_T(GPSIQ_SC08, 4); else if (slots == 8) GPSIQ_LAUNCH_T(GPSIQ_SC08, 8);
            else if (slots == 12) GPSIQ_LAUNCH_T(GPSIQ_SC08, 12); else GPSIQ_LAUNCH_T(GPSIQ_SC08, 16);
        }
#undef GPSIQ_LAUNCH_T
#undef GPSIQ_LAUNCH_T4
    } else if (variant == kRowsX) {
        const int tiles = (nsamp + kRowsTile - 1) / kRowsTile;
        dim3 grid((unsigned) (tiles * nblocks)), block(kRowsThreads);
#define GPSIQ_LAUNCH_X(F, N) hipLaunchKernelGGL((synth_rowsx<F, N>), grid, block, 0, stream, desc, nchan, nsamp, d, block_stride, block0, tab, tiles)
        const int slots = max_active <= 4 ? 4 : max_active <= 8 ? 8 : 16;
        if (sample_size == GPSIQ_SC16) {
            if (slots == 4) GPSIQ_LAUNCH_X(GPSIQ_SC16, 4); else if (slots == 8) GPSIQ_LAUNCH_X(GPSIQ_SC16, 8); else GPSIQ_LAUNCH_X(GPSIQ_SC16, 16);
        } else {
            if (slots == 4) GPSIQ_LAUNCH_X(GPSIQ_SC08, 4); else if (slots == 8) GPSIQ_LAUNCH_X(GPSIQ_SC08, 8); else GPSIQ_LAUNCH_X(GPSIQ_SC08, 16);
        }
#undef GPSIQ_LAUNCH_X
    } else if (variant == kRows) {
        const int tiles = (nsamp + kRowsTile - 1) / kRowsTile;
        dim3 grid((unsigned) (tiles * nblocks)), block(kRowsThreads);
        if (sample_size == GPSIQ_SC16)
            hipLaunchKernelGGL(synth_rows<GPSIQ_SC16>, grid, block, 0, stream, desc, nchan, nsamp, d, block_stride, block0, tab, tiles);
        else
            hipLaunchKernelGGL(synth_rows<GPSIQ_SC08>, grid, block, 0, stream, desc, nchan, nsamp, d, block_stride, block0, tab, tiles);
    } else {
        const int tile_samples = 4096;
        const int tiles = (nsamp + tile_samples - 1) / tile_samples;
        dim3 grid((unsigned) (tiles * nblocks)), block(kGenericThreads);
        if (sample_size == GPSIQ_SC16)
            hipLaunchKernelGGL(synth_generic<GPSIQ_SC16>, grid, block, 0, stream, desc, nchan, nsamp, d, block_stride, block0, tab, tiles, tile_samples);
        else
            hipLaunchKernelGGL(synth_generic<GPSIQ_SC08>, grid, block, 0, stream, desc, nchan, nsamp, d, block_stride, block0, tab, tiles, tile_samples);
    }
    return hipGetLastError();
}

}  // namespace gpsiq
