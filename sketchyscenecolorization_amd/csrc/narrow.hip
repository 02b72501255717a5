// narrow.hip -- convolutions with very few output channels (<= 4): the generator's last transposed conv
// (128 -> 3, models_collection.py:529-534), the PatchGAN logit conv (512 -> 1, :834-835) and the data
// gradient of the discriminator's first conv w.r.t. the 3 generated channels.
//
// On the MFMA tile kernel these waste >90 % of a 32-column tile (measured 1.4-7 TFLOP/s).  Here they are a
// direct convolution on the vector ALUs: a workgroup stages a (16+halo)^2 input patch in LDS ONCE per 32-channel
// chunk -- the folded norm + activation is applied once per input element instead of once per tap -- together
// with the filter slice, and each lane owns one lattice point (all 4 sub-pixel phases of it for the transposed
// form), reading the patch with conflict-free ds_read_b128 and the filter as LDS broadcasts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "host_util.h"

#define NT 16            // lattice tile edge (16x16 = 256 lanes)
#define NCH 32           // channels per chunk
#define NPAD 36          // floats per patch pixel (32 + 4: odd multiple of 16 B -> conflict-free b128 reads)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float nr_act(float v, int act) {
    if (act == SSC_ACT_RELU) return fmaxf(v, 0.f);
    if (act == SSC_ACT_LRELU) return fmaxf(v, 0.2f * v);
    return v;
}

// MODE 0: conv form, nphase == 1, in_stride == 1.   MODE 1: k=4 s=2 transposed form (4 phases).
// BIGK (conv form): filters up to 7x7 (the MRU generator's last conv, 64 -> 3 at 7x7, models_collection.py:372-374): a
// (16+6)^2 patch and 49 taps of filter per chunk -- more staging registers and LDS, one workgroup per CU.
template <int MODE, int NOUT, bool BIGK>
__global__ __launch_bounds__(256) void narrow_fwd_kernel(const ssc_conv_desc d, float* __restrict__ slabs,
                                                         long slab_stride, int csplit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = d.x.C0 + d.x.C1;
    const int PYD = (MODE == 1) ? NT + 2 : (NT - 1) + d.TH;
    const int PXD = (MODE == 1) ? NT + 2 : (NT - 1) + d.TW;
    const int NTAP = d.KH * d.KW;
    float* patch = smem;                         // [PYD*PXD][NPAD]
    float* wl = smem + PYD * PXD * NPAD;         // [NTAP][NOUT][NCH]

    const int tid = threadIdx.x;
    const int ly = tid >> 4, lx = tid & 15;
    // csplit > 1: the channel chunks are divided among csplit workgroups per tile (few lattice tiles, many channels: the
    // PatchGAN logit conv is 128 tiles x 512 channels); each writes its raw partial sums to slab `cslice`
    const int nimg = blockIdx.z / csplit;
    const int cslice = blockIdx.z - nimg * csplit;
    const int py0 = blockIdx.y * NT, px0 = blockIdx.x * NT;
    // input coordinate of patch element (0,0)
    const int iy0 = (MODE == 1) ? py0 - 1 : py0 + d.ioff_y;
    const int ix0 = (MODE == 1) ? px0 - 1 : px0 + d.ioff_x;

    // accumulators are float2 (even / odd channel pairs) so that the compiler emits packed v_pk_fma_f32:
    // two FMAs per vector instruction
    constexpr int NPH = (MODE == 1) ? 4 : 1;
    f32x2 acc[NPH][NOUT];
#pragma unroll
    for (int p = 0; p < NPH; ++p)
#pragma unroll
        for (int n = 0; n < NOUT; ++n) acc[p][n] = (f32x2){0.f, 0.f};

    // staging registers: all global loads of a chunk are issued before the first one is consumed
    constexpr int MAXE = BIGK ? 16 : 12;        // ceil(19*19*8 / 256) (22*22*8 / 256) patch float4 per thread
    constexpr int MAXW = ((BIGK ? 49 : 16) * NOUT * NCH + 255) / 256;      // filter floats per thread (<= 16 / 49 taps)
    const int c4s = (tid & 7) * 4;      // fixed per thread: 256 % 8 == 0
    float4 rv[MAXE];
    bool ok[MAXE];
    float4 ra = make_float4(1.f, 1.f, 1.f, 1.f), rb = make_float4(0.f, 0.f, 0.f, 0.f);
    float rw[MAXW];
    const int nwl = NTAP * NOUT * NCH;
    int cur_act = d.x.act;
    auto load_chunk = [&](int cb) {
        const bool first = cb < d.x.C0;
        const float* src = first ? d.x.s0 : d.x.s1;
        const int cs = first ? d.x.C0 : d.x.C1;
        const int cc = first ? cb : cb - d.x.C0;
        const float* abp = first ? d.x.ab0 : d.x.ab1;
        cur_act = (!first && d.x.act1 >= 0) ? d.x.act1 : d.x.act;
        ra = make_float4(1.f, 1.f, 1.f, 1.f);
        rb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (abp != nullptr) {
            ra = *reinterpret_cast<const float4*>(abp + cc + c4s);
            rb = *reinterpret_cast<const float4*>(abp + cs + cc + c4s);
        }
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int pos = (tid >> 3) + 32 * q;
            const int pr = pos / PXD, pc = pos - pr * PXD;
            const int iy = iy0 + pr, ix = ix0 + pc;
            ok[q] = pos < PYD * PXD && (unsigned)iy < (unsigned)d.x.H && (unsigned)ix < (unsigned)d.x.W;
            const long off = ok[q] ? (((long)nimg * d.x.H + iy) * d.x.W + ix) * cs : 0;
            rv[q] = *reinterpret_cast<const float4*>(src + off + cc + c4s);
        }
#pragma unroll
        for (int q = 0; q < MAXW; ++q) {
            const int e = tid + 256 * q;
            const int k = e & (NCH - 1);
            const int n = (e / NCH) % NOUT;
            const int tap = e / (NCH * NOUT);
            const bool v = e < nwl && n < d.Nn && cb + k < d.k_real;
            const long idx = (d.bmode == 0) ? ((long)tap * d.wC0 + cb + k) * d.wC1 + d.n_off + n
                                            : ((long)tap * d.wC0 + d.n_off + n) * d.wC1 + cb + k;
            const float w = d.w[v ? idx : 0];
            rw[q] = v ? w : 0.f;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int pos = (tid >> 3) + 32 * q;
            if (pos < PYD * PXD) {
                float4 v = rv[q];
                v.x = nr_act(fmaf(ra.x, v.x, rb.x), cur_act); v.y = nr_act(fmaf(ra.y, v.y, rb.y), cur_act);
                v.z = nr_act(fmaf(ra.z, v.z, rb.z), cur_act); v.w = nr_act(fmaf(ra.w, v.w, rb.w), cur_act);
                if (!ok[q]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(patch + pos * NPAD + c4s) = v;
            }
        }
#pragma unroll
        for (int q = 0; q < MAXW; ++q) {
            const int e = tid + 256 * q;
            if (e < nwl) wl[e] = rw[q];
        }
    };

    const int nchunk = C / NCH;
    const int cper = (nchunk + csplit - 1) / csplit;
    const int cb_begin = cslice * cper * NCH, cb_end = min(C, cb_begin + cper * NCH);
    for (int cb = cb_begin; cb < cb_end; cb += NCH) {
        // (prefetching the next chunk across the accumulate phase was measured slower: the 48 staging registers
        // held live across it cost more occupancy than the hidden latency buys)
        load_chunk(cb);
        store_chunk();
        __syncthreads();
        // ---- accumulate ----
        if (MODE == 1) {
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const int ry = ph >> 1, rx = ph & 1;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ty = t >> 1, tx = t & 1;
                    const float* xp = patch + ((ly + ry + ty) * PXD + lx + rx + tx) * NPAD;
                    const float* wp = wl + ((3 - ry - 2 * ty) * 4 + (3 - rx - 2 * tx)) * NOUT * NCH;
#pragma unroll
                    for (int c4 = 0; c4 < NCH; c4 += 4) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(xp + c4);
#pragma unroll
                        for (int n = 0; n < NOUT; ++n) {
                            const f32x4 w = *reinterpret_cast<const f32x4*>(wp + n * NCH + c4);
                            acc[ph][n] = __builtin_elementwise_fma(x.xy, w.xy, acc[ph][n]);
                            acc[ph][n] = __builtin_elementwise_fma(x.zw, w.zw, acc[ph][n]);
                        }
                    }
                }
            }
        } else {
            for (int ty = 0; ty < d.TH; ++ty) {
                for (int tx = 0; tx < d.TW; ++tx) {
                    const float* xp = patch + ((ly + ty) * PXD + lx + tx) * NPAD;
                    const int ky = d.ky0 + ty * d.kstep, kx = d.kx0 + tx * d.kstep;
                    const float* wp = wl + (ky * d.KW + kx) * NOUT * NCH;
#pragma unroll
                    for (int c4 = 0; c4 < NCH; c4 += 4) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(xp + c4);
#pragma unroll
                        for (int n = 0; n < NOUT; ++n) {
                            const f32x4 w = *reinterpret_cast<const f32x4*>(wp + n * NCH + c4);
                            acc[0][n] = __builtin_elementwise_fma(x.xy, w.xy, acc[0][n]);
                            acc[0][n] = __builtin_elementwise_fma(x.zw, w.zw, acc[0][n]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue ----
    const int py = py0 + ly, px = px0 + lx;
    if (py >= d.PH || px >= d.PW) return;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        const int oy = (MODE == 1) ? 2 * py + (ph >> 1) : py * d.out_stride + d.ooff_y;
        const int ox = (MODE == 1) ? 2 * px + (ph & 1) : px * d.out_stride + d.ooff_x;
        const long opix = (((long)nimg * d.OH + oy) * d.OW + ox) * d.ldc;
        float* o = (csplit > 1) ? slabs + (long)cslice * slab_stride + opix : d.out + opix;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (n >= d.Nstore) break;
            float v = 0.f;          // columns in [Nn, Nstore) are channel padding: written as 0
            if (n < NOUT && n < d.Nn) {
                v = acc[ph][n < NOUT ? n : 0].x + acc[ph][n < NOUT ? n : 0].y;
                if (csplit > 1) {   // bias, activation and accumulation belong to the reduce pass
                    o[n] = v;
                    continue;
                }
                if (d.bias != nullptr) v += d.bias[n];
                if (d.epi == 1) v = tanhf(v);
                else if (d.epi == 2) v = fmaxf(v, 0.2f * v);
                if (d.accumulate) v += o[n];
            }
            o[n] = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Scalar-filter form.  With the filter slice in LDS a lane issues one ds_read_b128 per two packed FMAs and, worse, a chunk is
// load -> LDS -> barrier -> accumulate -> barrier with nothing in flight during the accumulate.  Here:
//   * the filter is read with scalar loads into SGPR pairs (v_pk_fma_f32 takes one as an operand): no filter staging, no
//     filter LDS.  Needs the channels of one (tap, output) contiguous in memory: the [n][k] orientation or one output column;
//   * 16-channel chunks: 6 staging float4 per thread, so the global loads of chunk c+1 stay in flight across the accumulate
//     of chunk c, and two LDS images (2 x 26 KB, 3 workgroups per CU) need one barrier per chunk.
// ---------------------------------------------------------------------------------------------
#define SCH 16           // channels per chunk
#define SPAD 20          // floats per patch pixel (80 B: odd multiple of 16 B -> conflict-free b128 reads)
typedef const __attribute__((address_space(4))) float* cfloatp;
typedef const __attribute__((address_space(4))) f32x4* cf32x4p;

// KS: 0 = taps from the descriptor (<= 4x4); 7 = the 7x7 conv form with the tap loops unrolled (the MRU generator's last conv,
// 64 -> 3, models_collection.py:372-374: 3.2 ms per launch on narrow_fwd_kernel<BIGK> -- one 88 KB workgroup per CU, every
// filter value an LDS read -- against ~0.8 ms here: a (16+6)^2 patch of 16 channels per image, two images = 77 KB).
template <int MODE, int NOUT, int KS>
__global__ __launch_bounds__(256) void narrow_sc_kernel(const ssc_conv_desc d, float* __restrict__ slabs,
                                                        long slab_stride, int csplit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = d.x.C0 + d.x.C1;
    const int PYD = (MODE == 1) ? NT + 2 : (NT - 1) + d.TH;
    const int PXD = (MODE == 1) ? NT + 2 : (NT - 1) + d.TW;
    const int PSZ = PYD * PXD * SPAD;            // floats per patch image

    const int tid = threadIdx.x;
    const int ly = tid >> 4, lx = tid & 15;
    const int nimg = blockIdx.z / csplit;
    const int cslice = blockIdx.z - nimg * csplit;
    const int py0 = blockIdx.y * NT, px0 = blockIdx.x * NT;
    const int iy0 = (MODE == 1) ? py0 - 1 : py0 + d.ioff_y;
    const int ix0 = (MODE == 1) ? px0 - 1 : px0 + d.ioff_x;

    constexpr int NPH = (MODE == 1) ? 4 : 1;
    f32x2 acc[NPH][NOUT];
#pragma unroll
    for (int p = 0; p < NPH; ++p)
#pragma unroll
        for (int n = 0; n < NOUT; ++n) acc[p][n] = (f32x2){0.f, 0.f};

    // element (tap, n, channel k) of the filter at wsc + tap * w_ts + n * w_ns + k
    const long w_ts = (long)d.wC0 * d.wC1;
    const long w_ns = (d.bmode == 0) ? 0 : (long)d.wC1;
    const cfloatp wsc = (cfloatp)(d.w + ((d.bmode == 0) ? (long)d.n_off : (long)d.n_off * d.wC1));

    constexpr int MAXE = (KS == 7) ? 8 : 6;     // ceil(19 * 19 * 4 / 256) (22 * 22 * 4 / 256) patch float4 per thread
    const int c4s = (tid & 3) * 4;
    // per-thread patch positions: fixed for the whole kernel
    int poff[MAXE];                     // pixel index into the source, or -1: outside the image / the patch
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        const int pos = (tid >> 2) + 64 * q;
        const int pr = pos / PXD, pc = pos - pr * PXD;
        const int iy = iy0 + pr, ix = ix0 + pc;
        const bool ok = pos < PYD * PXD && (unsigned)iy < (unsigned)d.x.H && (unsigned)ix < (unsigned)d.x.W;
        poff[q] = ok ? (nimg * d.x.H + iy) * d.x.W + ix : -1;
    }
    float4 rv[MAXE];
    float4 ra, rb;
    int cur_act = d.x.act;
    auto load_chunk = [&](int cb) {
        const bool first = cb < d.x.C0;
        const float* src = first ? d.x.s0 : d.x.s1;
        const int cs = first ? d.x.C0 : d.x.C1;
        const int cc = first ? cb : cb - d.x.C0;
        const float* abp = first ? d.x.ab0 : d.x.ab1;
        cur_act = (!first && d.x.act1 >= 0) ? d.x.act1 : d.x.act;
        ra = make_float4(1.f, 1.f, 1.f, 1.f);
        rb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (abp != nullptr) {
            ra = *reinterpret_cast<const float4*>(abp + cc + c4s);
            rb = *reinterpret_cast<const float4*>(abp + cs + cc + c4s);
        }
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const long off = poff[q] >= 0 ? (long)poff[q] * cs : 0;
            rv[q] = *reinterpret_cast<const float4*>(src + off + cc + c4s);
        }
    };
    auto store_chunk = [&](float* patch) {
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int pos = (tid >> 2) + 64 * q;
            if (pos < PYD * PXD) {
                float4 v = rv[q];
                v.x = nr_act(fmaf(ra.x, v.x, rb.x), cur_act); v.y = nr_act(fmaf(ra.y, v.y, rb.y), cur_act);
                v.z = nr_act(fmaf(ra.z, v.z, rb.z), cur_act); v.w = nr_act(fmaf(ra.w, v.w, rb.w), cur_act);
                if (poff[q] < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(patch + pos * SPAD + c4s) = v;
            }
        }
    };

    const int nchunk = C / SCH;
    const int cper = (nchunk + csplit - 1) / csplit;
    const int cb_begin = cslice * cper * SCH, cb_end = min(C, cb_begin + cper * SCH);
    if (cb_begin < cb_end) load_chunk(cb_begin);
    int buf = 0;
    for (int cb = cb_begin; cb < cb_end; cb += SCH) {
        float* patch = smem + buf * PSZ;
        store_chunk(patch);
        // one barrier per chunk: the image written next (the other one) was last read before this barrier
        __syncthreads();
        if (cb + SCH < cb_end) load_chunk(cb + SCH);        // in flight across the accumulate below
        if (MODE == 1) {
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const int ry = ph >> 1, rx = ph & 1;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ty = t >> 1, tx = t & 1;
                    const float* xp = patch + ((ly + ry + ty) * PXD + lx + rx + tx) * SPAD;
                    const cfloatp ws = wsc + ((3 - ry - 2 * ty) * 4 + (3 - rx - 2 * tx)) * w_ts + cb;
#pragma unroll
                    for (int c4 = 0; c4 < SCH; c4 += 4) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(xp + c4);
#pragma unroll
                        for (int n = 0; n < NOUT; ++n) {
                            const f32x4 w = *(cf32x4p)(ws + n * w_ns + c4);
                            acc[ph][n] = __builtin_elementwise_fma(x.xy, w.xy, acc[ph][n]);
                            acc[ph][n] = __builtin_elementwise_fma(x.zw, w.zw, acc[ph][n]);
                        }
                    }
                }
            }
        } else if (KS == 7) {       // unflipped 7x7 taps (ky0 = kx0 = 0, kstep = 1: the host checks), fully unrolled
#pragma unroll
            for (int ty = 0; ty < 7; ++ty) {
#pragma unroll
                for (int tx = 0; tx < 7; ++tx) {
                    const float* xp = patch + ((ly + ty) * PXD + lx + tx) * SPAD;
                    const cfloatp ws = wsc + (ty * 7 + tx) * w_ts + cb;
#pragma unroll
                    for (int c4 = 0; c4 < SCH; c4 += 4) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(xp + c4);
#pragma unroll
                        for (int n = 0; n < NOUT; ++n) {
                            const f32x4 w = *(cf32x4p)(ws + n * w_ns + c4);
                            acc[0][n] = __builtin_elementwise_fma(x.xy, w.xy, acc[0][n]);
                            acc[0][n] = __builtin_elementwise_fma(x.zw, w.zw, acc[0][n]);
                        }
                    }
                }
            }
        } else {
            for (int ty = 0; ty < d.TH; ++ty) {
                for (int tx = 0; tx < d.TW; ++tx) {
                    const float* xp = patch + ((ly + ty) * PXD + lx + tx) * SPAD;
                    const int ky = d.ky0 + ty * d.kstep, kx = d.kx0 + tx * d.kstep;
                    const cfloatp ws = wsc + (ky * d.KW + kx) * w_ts + cb;
#pragma unroll
                    for (int c4 = 0; c4 < SCH; c4 += 4) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(xp + c4);
#pragma unroll
                        for (int n = 0; n < NOUT; ++n) {
                            const f32x4 w = *(cf32x4p)(ws + n * w_ns + c4);
                            acc[0][n] = __builtin_elementwise_fma(x.xy, w.xy, acc[0][n]);
                            acc[0][n] = __builtin_elementwise_fma(x.zw, w.zw, acc[0][n]);
                        }
                    }
                }
            }
        }
        buf ^= 1;
    }

    // ---- epilogue (as narrow_fwd_kernel) ----
    const int py = py0 + ly, px = px0 + lx;
    if (py >= d.PH || px >= d.PW) return;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        const int oy = (MODE == 1) ? 2 * py + (ph >> 1) : py * d.out_stride + d.ooff_y;
        const int ox = (MODE == 1) ? 2 * px + (ph & 1) : px * d.out_stride + d.ooff_x;
        const long opix = (((long)nimg * d.OH + oy) * d.OW + ox) * d.ldc;
        float* o = (csplit > 1) ? slabs + (long)cslice * slab_stride + opix : d.out + opix;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (n >= d.Nstore) break;
            float v = 0.f;
            if (n < NOUT && n < d.Nn) {
                v = acc[ph][n < NOUT ? n : 0].x + acc[ph][n < NOUT ? n : 0].y;
                if (csplit > 1) {
                    o[n] = v;
                    continue;
                }
                if (d.bias != nullptr) v += d.bias[n];
                if (d.epi == 1) v = tanhf(v);
                else if (d.epi == 2) v = fmaxf(v, 0.2f * v);
                if (d.accumulate) v += o[n];
            }
            o[n] = v;
        }
    }
}

// the scalar-filter form applies: channels of a (tap, output) contiguous and 16-byte aligned, <= 4x4 taps
static bool narrow_sc_ok(const ssc_conv_desc& d) {
    static const bool on = ssc_dev_getenv("SSC_NARROW_WSC") == nullptr || atoi(ssc_dev_getenv("SSC_NARROW_WSC")) != 0;
    const int nout = d.Nn < 1 ? 1 : d.Nn;
    const bool kcontig = (d.bmode == 1) ? (d.wC1 % 4 == 0) : (d.wC1 == 1 && nout == 1 && d.n_off == 0);
    if (!on || !kcontig || (reinterpret_cast<uintptr_t>(d.w) & 15) != 0) return false;
    // measured (scripts/conv_microbench.py, batch 32): 128 -> 3 transposed 135 -> 127 us, the 64 -> 3 data gradient 79 -> 59 us,
    // but the 512 -> 1 conv form 41 -> 56 us (one output: a scalar fetch feeds a single packed FMA) -- transposed form only.
    // What bounds it now is the scalar data path: with constant weights the same launch takes 83 us, without the accumulate 49.
    // Round 4, two lattice points per lane (32 x 16 tiles, each patch pixel read once per point, 8-channel chunks): with the
    // filter slice in LDS read as broadcasts 17.89 vs 17.78 ms per train step, with scalar filter loads (one fetch per 4 packed
    // FMAs) 291 vs 127 us for this layer and 115 vs 59 us for the 64 -> 3 data gradient -- the LDS -> VGPR return path
    // (128 B / clk per CU, broadcast or not) bounds the first, 134 VGPRs + the lgkmcnt shared by scalar and LDS returns the
    // second.  Not kept (profiles/NOTEBOOK_r04.md).
    // ... and the unflipped 7x7 conv form with >= 2 outputs and an [n][k] filter (hip.conv_forward(..., w_nk=...): the MRU
    // generator's last conv hands over a transposed copy of its [7,7,64,3] filter)
    const bool conv7 = d.nphase == 1 && d.bmode == 1 && nout >= 2 && d.TH == 7 && d.TW == 7 && d.KH == 7 && d.KW == 7 &&
                       d.ky0 == 0 && d.kx0 == 0 && d.kstep == 1 && ((d.x.C0 + d.x.C1) % SCH) == 0 && (d.x.C0 % SCH) == 0;
    if (d.nphase != 4 && !conv7) return false;
    // pixel offsets are 32-bit in this form
    return (long)d.NB * d.x.H * d.x.W < 0x7fffffffL;
}

extern "C" int ssc_conv_narrow_supported(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    const int C = d.x.C0 + d.x.C1;
    if (d.Nn > 4 || d.Nstore > 4) return 0;
    if ((C % NCH) != 0 || (d.x.C0 % NCH) != 0 || d.k_real != C) return 0;
    if (d.nphase == 4) return (d.TH == 2 && d.TW == 2 && d.KH == 4 && d.KW == 4) ? 1 : 0;
    if (d.nphase != 1 || d.in_stride != 1 || d.TH > 7 || d.TW > 7 || d.KH * d.KW > 49) return 0;
    return 1;
}

template <int MODE>
static int launch_narrow(const ssc_conv_desc& d, hipStream_t st, float* ws, int64_t ws_bytes, int* csplit_out) {
    const int PYD = (MODE == 1) ? NT + 2 : (NT - 1) + d.TH;
    const int PXD = (MODE == 1) ? NT + 2 : (NT - 1) + d.TW;
    const int nout = d.Nn < 1 ? 1 : d.Nn;       // accumulators per phase (1..4)
    const bool sc = narrow_sc_ok(d);
    const size_t lds = sc ? (size_t)2 * PYD * PXD * SPAD * sizeof(float)
                          : ((size_t)PYD * PXD * NPAD + (size_t)d.KH * d.KW * nout * NCH) * sizeof(float);
    dim3 grid((d.PW + NT - 1) / NT, (d.PH + NT - 1) / NT, d.NB);
    // channel split when the lattice alone gives the chip too few workgroups (conv form only)
    int csplit = 1;
    const long out_count = (long)d.NB * d.OH * d.OW * d.ldc;
    if (MODE == 0 && ws != nullptr && csplit_out != nullptr) {
        const long wgs = (long)grid.x * grid.y * grid.z;
        const int nchunk = (d.x.C0 + d.x.C1) / NCH;
        while (csplit < 8 && wgs * csplit < 512 && nchunk / (csplit * 2) >= 2 &&
               (int64_t)(csplit * 2) * out_count * 4 <= ws_bytes)
            csplit *= 2;
    }
    if (csplit_out != nullptr) *csplit_out = csplit;
    grid.z *= csplit;
#define NARROW_LAUNCH(NO)                                                                                          \
    {                                                                                                              \
        static unsigned long long attr_done = 0;                                                                   \
        const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&narrow_fwd_kernel<MODE, NO, BIG>), 96 * 1024, \
                                        &attr_done);                                                               \
        if (arc != 0) return arc;                                                                                  \
        hipLaunchKernelGGL((narrow_fwd_kernel<MODE, NO, BIG>), grid, dim3(256), lds, st, d, ws, out_count, csplit); \
    }
#define NARROW_SC_LAUNCH(NO) hipLaunchKernelGGL((narrow_sc_kernel<1, NO, 0>), grid, dim3(256), lds, st, d, ws, out_count, csplit);
#define NARROW_SC7_LAUNCH(NO)                                                                                      \
    {                                                                                                              \
        static unsigned long long attr7_done = 0;                                                                  \
        const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&narrow_sc_kernel<0, NO, 7>), 96 * 1024,     \
                                        &attr7_done);                                                              \
        if (arc != 0) return arc;                                                                                  \
        hipLaunchKernelGGL((narrow_sc_kernel<0, NO, 7>), grid, dim3(256), lds, st, d, ws, out_count, csplit);      \
    }
    if (lds > 96 * 1024) return -5;
    if (sc && MODE == 0) {      // the 7x7 conv form (narrow_sc_ok): no channel split
        if (csplit_out != nullptr) *csplit_out = 1;
        grid.z = d.NB;
        csplit = 1;
        if (nout == 2) NARROW_SC7_LAUNCH(2) else if (nout == 3) NARROW_SC7_LAUNCH(3) else NARROW_SC7_LAUNCH(4)
    } else if (sc) {            // transposed form
        if (nout == 1) NARROW_SC_LAUNCH(1) else if (nout == 2) NARROW_SC_LAUNCH(2) else if (nout == 3) NARROW_SC_LAUNCH(3) else NARROW_SC_LAUNCH(4)
    } else if (MODE == 0 && (d.TH > 4 || d.TW > 4 || d.KH * d.KW > 16)) {
        constexpr bool BIG = true;
        if (nout == 1) NARROW_LAUNCH(1) else if (nout == 2) NARROW_LAUNCH(2) else if (nout == 3) NARROW_LAUNCH(3) else NARROW_LAUNCH(4)
    } else {
        constexpr bool BIG = false;
        if (nout == 1) NARROW_LAUNCH(1) else if (nout == 2) NARROW_LAUNCH(2) else if (nout == 3) NARROW_LAUNCH(3) else NARROW_LAUNCH(4)
    }
#undef NARROW_LAUNCH
#undef NARROW_SC_LAUNCH
#undef NARROW_SC7_LAUNCH
    return (int)hipGetLastError();
}

extern "C" int ssc_conv_narrow_forward(const ssc_conv_desc* dp, void* stream) {
    if (!ssc_conv_narrow_supported(dp)) return -1;
    return dp->nphase == 4 ? launch_narrow<1>(*dp, (hipStream_t)stream, nullptr, 0, nullptr)
                           : launch_narrow<0>(*dp, (hipStream_t)stream, nullptr, 0, nullptr);
}

// with a workspace: may split the channels (csplit_out > 1); the caller then sums the csplit slabs of the workspace
// (slab stride = NB*OH*OW*ldc floats) and applies bias / activation / accumulation (slab_reduce_kernel in igemm.hip)
int ssc_conv_narrow_forward_ws(const ssc_conv_desc* dp, float* ws, int64_t ws_bytes, void* stream, int* csplit_out) {
    if (!ssc_conv_narrow_supported(dp)) return -1;
    return dp->nphase == 4 ? launch_narrow<1>(*dp, (hipStream_t)stream, ws, ws_bytes, csplit_out)
                           : launch_narrow<0>(*dp, (hipStream_t)stream, ws, ws_bytes, csplit_out);
}
