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
#include "sketchycolor_hip.h"

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
    const size_t lds = ((size_t)PYD * PXD * NPAD + (size_t)d.KH * d.KW * nout * NCH) * sizeof(float);
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
        static bool attr = false;                                                                                  \
        if (!attr) {                                                                                               \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&narrow_fwd_kernel<MODE, NO, BIG>),            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                      \
            attr = true;                                                                                           \
        }                                                                                                          \
        hipLaunchKernelGGL((narrow_fwd_kernel<MODE, NO, BIG>), grid, dim3(256), lds, st, d, ws, out_count, csplit); \
    }
    if (lds > 96 * 1024) return -5;
    if (MODE == 0 && (d.TH > 4 || d.TW > 4 || d.KH * d.KW > 16)) {
        constexpr bool BIG = true;
        if (nout == 1) NARROW_LAUNCH(1) else if (nout == 2) NARROW_LAUNCH(2) else if (nout == 3) NARROW_LAUNCH(3) else NARROW_LAUNCH(4)
    } else {
        constexpr bool BIG = false;
        if (nout == 1) NARROW_LAUNCH(1) else if (nout == 2) NARROW_LAUNCH(2) else if (nout == 3) NARROW_LAUNCH(3) else NARROW_LAUNCH(4)
    }
#undef NARROW_LAUNCH
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
