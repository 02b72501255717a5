// losses_optim.hip -- loss reductions (graph_single.py:317-581, live branch), the
// spectral-norm power iteration (sn.py:12-52), TF-style Adam (graph_single.py:588)
// and flat-buffer helpers.  Everything here is HBM- or latency-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sketchycolor_hip.h"

#define CHECK_LAUNCH() ((int)hipGetLastError())

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum (256 threads); result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum_f(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

__device__ __forceinline__ float softplusf_(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf2_(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------ patch GAN loss (get_loss_wgan_sn, :388-419)
// loss_acc[slot] += scale * sum_r softplus(sign*x[r*ld]);  grad[r*ld] = gscale*sign*sigmoid(sign*x)
__global__ __launch_bounds__(256) void softplus_loss_kernel(const float* __restrict__ x, int ld, long rows, float sign,
                                                             float scale, double* __restrict__ loss_acc,
                                                             float* __restrict__ grad, float gscale) {
    __shared__ float sh[4];
    float s = 0.f;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        const float v = sign * x[r * ld];
        s += softplusf_(v);
        if (grad != nullptr) grad[r * ld] = gscale * sign * sigmoidf2_(v);
    }
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) atomicAdd(loss_acc, (double)scale * (double)t);
}

extern "C" int ssc_softplus_loss(const float* x, int ld, int64_t rows, float sign, float scale, double* loss_acc,
                                 float* grad, float gscale, void* stream) {
    long blocks = (rows + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(softplus_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ld,
                       (long)rows, sign, scale, loss_acc, grad, gscale);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ auxiliary-classifier loss (get_acgan_loss_focal, :340-353)
// focal=0: loss += coef*mean CE;  focal=1: loss += coef*mean (1-p_t)^2 * CE.  One wavefront per sample, K <= 64.
__global__ __launch_bounds__(64) void acgan_loss_kernel(const float* __restrict__ logits, const int* __restrict__ labels,
                                                         int N, int K, int focal, float coef,
                                                         double* __restrict__ loss_acc, float* __restrict__ dlogits) {
    const int n = blockIdx.x;
    const int lane = threadIdx.x;
    const float v = lane < K ? logits[(long)n * K + lane] : -INFINITY;
    float m = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float e = lane < K ? expf(v - m) : 0.f;
    const float se = wave_sum_f(e);
    const float p = e / se;
    const int t = min(max(labels[n], 0), K - 1);     // a label outside [0, K) must not select an invalid lane
    const float pt = __shfl(p, t, 64);
    const float vt = __shfl(v, t, 64);
    const float ce = -(vt - m - logf(se));
    float loss, g;
    if (focal) {
        loss = (1.f - pt) * (1.f - pt) * ce;
        // d/dp_t [(1-p)^2 * (-log p)] = 2(1-p) log p - (1-p)^2/p ; dp_t/dz_j = p_t(delta_tj - p_j).  The product
        // (dL/dp_t) * p_t is written out without the division: when p_t underflows to 0 (logit gap > ~88) the quotient
        // form is inf * 0 = NaN where TF's autodiff (through log_softmax) stays finite.
        const float fpp = -2.f * (1.f - pt) * ce * pt - (1.f - pt) * (1.f - pt);
        g = fpp * ((lane == t ? 1.f : 0.f) - p);
    } else {
        loss = ce;
        g = p - (lane == t ? 1.f : 0.f);
    }
    if (lane < K && dlogits != nullptr) dlogits[(long)n * K + lane] = coef * g / (float)N;
    if (lane == 0) atomicAdd(loss_acc, (double)coef * (double)loss / (double)N);
}

extern "C" int ssc_acgan_loss(const float* logits, const int* labels, int N, int K, int focal, float coef,
                              double* loss_acc, float* dlogits, void* stream) {
    if (K > 64) return -1;
    hipLaunchKernelGGL(acgan_loss_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, logits, labels, N, K, focal, coef,
                       loss_acc, dlogits);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ generator output: smooth-L1 + tanh backward
// gen[p*ldg + c], img[p*ldi + c], c < 3.  loss += coef*mean smoothL1(img-gen)   (graph_single.py:551-555)
// dpre[p*4 + c] = (gd[p*ldd + c] + coef/count * clamp(gen-img,-1,1)) * (1-gen^2);  dpre[p*4+3] = 0
__global__ __launch_bounds__(256) void gen_output_grad_kernel(const float* __restrict__ gen, int ldg,
                                                               const float* __restrict__ img, int ldi,
                                                               const float* __restrict__ gd, int ldd, long npix,
                                                               float coef, double* __restrict__ loss_acc,
                                                               float* __restrict__ dpre) {
    __shared__ float sh[4];
    const float inv = 1.f / (3.f * (float)npix);
    float s = 0.f;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        float* op = &o.x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = gen[p * ldg + c];
            const float d = g - img[p * ldi + c];
            const float a = fabsf(d);
            s += a < 1.f ? 0.5f * a * a : a - 0.5f;
            float dgen = coef * inv * fminf(fmaxf(d, -1.f), 1.f);
            if (gd != nullptr) dgen += gd[p * ldd + c];
            op[c] = dgen * (1.f - g * g);
        }
        if (dpre != nullptr) *reinterpret_cast<float4*>(dpre + p * 4) = o;
    }
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) atomicAdd(loss_acc, (double)coef * (double)inv * (double)t);
}

extern "C" int ssc_gen_output_grad(const float* gen, int ldg, const float* img, int ldi, const float* gd, int ldd,
                                   int64_t npix, float coef, double* loss_acc, float* dpre, void* stream) {
    long blocks = (npix + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(gen_output_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gen, ldg, img,
                       ldi, gd, ldd, (long)npix, coef, loss_acc, dpre);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ l2 regulariser (ly.l2_regularizer, mru.py:55,60)
// loss += rate*sum(w^2)/2 ; grad += rate*w
__global__ __launch_bounds__(256) void l2_reg_kernel(const float* __restrict__ w, long n, float rate,
                                                      double* __restrict__ loss_acc, float* __restrict__ grad) {
    __shared__ float sh[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = w[i];
        s += v * v;
        if (grad != nullptr) grad[i] += rate * v;
    }
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0 && loss_acc != nullptr) atomicAdd(loss_acc, 0.5 * (double)rate * (double)t);
}

extern "C" int ssc_l2_reg(const float* w, int64_t n, float rate, double* loss_acc, float* grad, void* stream) {
    long blocks = (n + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(l2_reg_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (long)n, rate,
                       loss_acc, grad);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ TF Adam (tf.train.AdamOptimizer dense apply)
// g = gscale*grad; m = b1*m + (1-b1)*g (m may be NULL when b1 == 0); v = b2*v + (1-b2)*g^2;
// var -= lr_t * m / (sqrt(v) + eps)      -- eps outside the bias-corrected sqrt, lr_t from the host
__global__ void adam_tf_kernel(float* __restrict__ var, const float* __restrict__ grad, float* __restrict__ m,
                               float* __restrict__ v, long n, float lr_t, const float* __restrict__ lr_dev, float b1,
                               float b2, float eps, float gscale) {
    if (lr_dev != nullptr) lr_t = *lr_dev;      // step size kept in device memory (graph replay)
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        const float4 g4 = *reinterpret_cast<const float4*>(grad + i);
        float4 v4 = *reinterpret_cast<const float4*>(v + i);
        float4 w4 = *reinterpret_cast<const float4*>(var + i);
        float g[4] = {g4.x * gscale, g4.y * gscale, g4.z * gscale, g4.w * gscale};
        float vv[4] = {v4.x, v4.y, v4.z, v4.w};
        float ww[4] = {w4.x, w4.y, w4.z, w4.w};
        float mm[4];
        if (m != nullptr) {
            const float4 m4 = *reinterpret_cast<const float4*>(m + i);
            mm[0] = m4.x; mm[1] = m4.y; mm[2] = m4.z; mm[3] = m4.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float mk = g[k] * (1.f - b1);
            if (m != nullptr) { mk = b1 * mm[k] + (1.f - b1) * g[k]; mm[k] = mk; }
            vv[k] = b2 * vv[k] + (1.f - b2) * g[k] * g[k];
            ww[k] -= lr_t * mk / (sqrtf(vv[k]) + eps);
        }
        *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        *reinterpret_cast<float4*>(var + i) = make_float4(ww[0], ww[1], ww[2], ww[3]);
        if (m != nullptr) *reinterpret_cast<float4*>(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
    }
}

extern "C" int ssc_adam_tf(float* var, const float* grad, float* m, float* v, int64_t n, float lr_t,
                           const float* lr_dev, float beta1, float beta2, float eps, float gscale, void* stream) {
    if (n & 3) return -1;   // flat parameter buffers are padded to 4 floats
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_tf_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, var, grad, m, v,
                       (long)n, lr_t, lr_dev, beta1, beta2, eps, gscale);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ the other optimizers of get_optimizer (graph_single.py:584-593)
// kind 1: tf.train.RMSPropOptimizer(lr, decay=h0, momentum=h1, epsilon=h2)  (slots: s1 = ms [init 1], s2 = mom [init 0])
//         ms += (g*g - ms)*(1-decay); mom = momentum*mom + lr*g*rsqrt(ms + eps); var -= mom
// kind 2: tf.train.AdagradOptimizer(lr)  (s1 = accumulator [init 0.1]):  acc += g*g; var -= lr*g*rsqrt(acc)
// kind 3: tf.train.AdadeltaOptimizer(lr, rho=h0, epsilon=h2)  (s1 = accum, s2 = accum_update [init 0]):
//         accum = rho*accum + (1-rho)*g*g; upd = sqrt(accum_update+eps)*rsqrt(accum+eps)*g;
//         accum_update = rho*accum_update + (1-rho)*upd*upd; var -= lr*upd
__global__ void optimizer_step_kernel(int kind, float* __restrict__ var, const float* __restrict__ grad,
                                      float* __restrict__ s1, float* __restrict__ s2, long n,
                                      const float* __restrict__ lr_dev, float h0, float h1, float h2, float gscale) {
    const float lr = *lr_dev;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float g = grad[i] * gscale;
        if (kind == 1) {
            const float ms = s1[i] + (g * g - s1[i]) * (1.f - h0);
            const float mom = h1 * s2[i] + lr * g * rsqrtf(ms + h2);
            s1[i] = ms;
            s2[i] = mom;
            var[i] -= mom;
        } else if (kind == 2) {
            const float acc = s1[i] + g * g;
            s1[i] = acc;
            var[i] -= lr * g * rsqrtf(acc);
        } else {
            const float acc = h0 * s1[i] + (1.f - h0) * g * g;
            const float upd = sqrtf(s2[i] + h2) * rsqrtf(acc + h2) * g;
            s1[i] = acc;
            s2[i] = h0 * s2[i] + (1.f - h0) * upd * upd;
            var[i] -= lr * upd;
        }
    }
}

extern "C" int ssc_optimizer_step(int kind, float* var, const float* grad, float* s1, float* s2, int64_t n,
                                  const float* lr_dev, float h0, float h1, float h2, float gscale, void* stream) {
    if (kind < 1 || kind > 3 || lr_dev == nullptr || s1 == nullptr || (kind != 2 && s2 == nullptr)) return -1;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(optimizer_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, kind, var, grad,
                       s1, s2, (long)n, lr_dev, h0, h1, h2, gscale);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ spectral norm (sn.py:12-52, num_iters = 1)
// W [m, n] row-major (n <= 64), u [n].  One workgroup.
//   a = u W^T; v = a/(|a|+eps); b = v W; u' = b/(|b|+eps); sigma = b.u'; Wbar = W/sigma
// aux = [sigma, |a|, |b|]
__global__ __launch_bounds__(256) void sn_forward_kernel(const float* __restrict__ W, const float* __restrict__ u, int m,
                                                          int n, float* __restrict__ v, float* __restrict__ u_new,
                                                          float* __restrict__ wbar, float* __restrict__ aux) {
    __shared__ float sh[4];
    __shared__ float bsh[64];
    __shared__ float scal[3];
    const int tid = threadIdx.x;
    float ss = 0.f;
    for (int i = tid; i < m; i += 256) {
        float a = 0.f;
        for (int j = 0; j < n; ++j) a += u[j] * W[(long)i * n + j];
        v[i] = a;           // un-normalised for now
        ss += a * a;
    }
    const float t = block_sum_256(ss, sh);
    if (tid == 0) scal[1] = sqrtf(t);
    if (tid < 64) bsh[tid] = 0.f;
    __syncthreads();
    const float ra = scal[1];
    const float inva = 1.f / (ra + 1e-12f);
    for (int i = tid; i < m; i += 256) v[i] *= inva;
    __syncthreads();
    // b[j] = sum_i v[i] W[i][j]  : wave w handles rows w, w+4, ...; lane = column; the four partial sums are
    // combined in a fixed order (no atomics: results are reproducible run to run)
    __shared__ float bpart[4][64];
    {
        const int lane = tid & 63, wave = tid >> 6;
        float acc = 0.f;
        if (lane < n)
            for (int i = wave; i < m; i += 4) acc += v[i] * W[(long)i * n + lane];
        bpart[wave][lane] = acc;
    }
    __syncthreads();
    if (tid < 64) bsh[tid] = (bpart[0][tid] + bpart[1][tid]) + (bpart[2][tid] + bpart[3][tid]);
    __syncthreads();
    float bb = (tid < n) ? bsh[tid] * bsh[tid] : 0.f;
    const float t2 = block_sum_256(bb, sh);
    if (tid == 0) {
        const float rb = sqrtf(t2);
        scal[2] = rb;
        scal[0] = t2 / (rb + 1e-12f);    // sigma = b . (b/(|b|+eps))
    }
    __syncthreads();
    const float rb = scal[2], sigma = scal[0];
    if (tid < n) u_new[tid] = bsh[tid] / (rb + 1e-12f);
    for (long i = tid; i < (long)m * n; i += 256) wbar[i] = W[i] / sigma;
    if (tid == 0) { aux[0] = sigma; aux[1] = ra; aux[2] = rb; }
}

// dW (+)= G/sigma + v^T gb + ga^T u      (gradient through sigma and the power iteration)
__global__ __launch_bounds__(256) void sn_backward_kernel(const float* __restrict__ W, const float* __restrict__ u,
                                                           const float* __restrict__ v,
                                                           const float* __restrict__ u_new,
                                                           const float* __restrict__ aux, const float* __restrict__ G,
                                                           int m, int n, float* __restrict__ dW, int accumulate,
                                                           float* __restrict__ scratch /* m floats */) {
    __shared__ float sh[4];
    __shared__ float gb[64];
    __shared__ float scal[2];
    const int tid = threadIdx.x;
    const float sigma = aux[0], ra = aux[1], rb = aux[2];
    const float eps = 1e-12f;
    float s = 0.f;
    for (long i = tid; i < (long)m * n; i += 256) s += G[i] * W[i];
    const float gw = block_sum_256(s, sh);
    if (tid == 0) scal[0] = -gw / (sigma * sigma);      // dL/dsigma
    __syncthreads();
    const float dsig = scal[0];
    // sigma = s^2/(s+eps), s=|b| ; b = u'(s+eps)
    const float dsds = (rb * rb + 2.f * rb * eps) / ((rb + eps) * (rb + eps));
    if (tid < n) {
        const float bj = u_new[tid] * (rb + eps);
        gb[tid] = dsig * dsds * bj / rb;
    }
    __syncthreads();
    // gv[i] = sum_j gb[j] W[i][j] ; gva = gv . a, a = v*(ra+eps)
    float dot = 0.f;
    for (int i = tid; i < m; i += 256) {
        float gvi = 0.f;
        for (int j = 0; j < n; ++j) gvi += gb[j] * W[(long)i * n + j];
        scratch[i] = gvi;
        dot += gvi * v[i] * (ra + eps);
    }
    const float gva = block_sum_256(dot, sh);
    if (tid == 0) scal[1] = gva;
    __syncthreads();
    const float gvdot = scal[1];
    for (long k = tid; k < (long)m * n; k += 256) {
        const int i = (int)(k / n), j = (int)(k - (long)i * n);
        const float ai = v[i] * (ra + eps);
        const float ga = scratch[i] / (ra + eps) - gvdot * ai / (ra * (ra + eps) * (ra + eps));
        float d = G[k] / sigma + v[i] * gb[j] + ga * u[j];
        if (accumulate) d += dW[k];
        dW[k] = d;
    }
}

extern "C" int ssc_sn_forward(const float* W, const float* u, int m, int n, float* v, float* u_new, float* wbar,
                              float* aux, void* stream) {
    if (n > 64) return -1;
    hipLaunchKernelGGL(sn_forward_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, W, u, m, n, v, u_new, wbar, aux);
    return CHECK_LAUNCH();
}

extern "C" int ssc_sn_backward(const float* W, const float* u, const float* v, const float* u_new, const float* aux,
                               const float* G, int m, int n, float* dW, int accumulate, float* scratch, void* stream) {
    if (n > 64) return -1;
    hipLaunchKernelGGL(sn_backward_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, W, u, v, u_new, aux, G, m, n, dW,
                       accumulate, scratch);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ spectral norm for weights of any size
// (every conv / FC weight of the MRU discriminator, mru.py:123, 68; sn.py:12-52).  Same math as the single-block
// kernels above, split into row-dot / column-dot / scalar stages so a [6912, 768] filter uses the whole chip.
// All reductions are two-stage with a fixed order (reproducible).
__global__ void sn_rowdot_kernel(const float* __restrict__ W, const float* __restrict__ x, int m, int n,
                                 float* __restrict__ out) {         // out[i] = sum_j x[j] W[i][j], one wavefront per row
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    float acc = 0.f;
    for (int j = lane; j < n; j += 64) acc += x[j] * W[(long)row * n + j];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) out[row] = acc;
}

__global__ void sn_coldot_partial_kernel(const float* __restrict__ W, const float* __restrict__ v, int m, int n,
                                         int nsplit, float* __restrict__ part) {   // part[s][j] = sum_{i in s} v[i] W[i][j]
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane, s = blockIdx.y;
    const int rows = (m + nsplit - 1) / nsplit;
    const int r0 = s * rows, r1 = min(m, r0 + rows);
    float acc = 0.f;
    if (j < n)
        for (int i = r0 + rl; i < r1; i += 4) acc += v[i] * W[(long)i * n + j];
    sh[rl][lane] = acc;
    __syncthreads();
    if (rl == 0 && j < n) part[(long)s * n + j] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}

__global__ __launch_bounds__(256) void sn_normalize_v_kernel(float* __restrict__ v, int m, float* __restrict__ aux) {
    __shared__ float sh[4];
    __shared__ float sc;
    float ss = 0.f;
    for (int i = threadIdx.x; i < m; i += 256) ss += v[i] * v[i];
    const float t = block_sum_256(ss, sh);
    if (threadIdx.x == 0) { sc = sqrtf(t); aux[1] = sc; }
    __syncthreads();
    const float inv = 1.f / (sc + 1e-12f);
    for (int i = threadIdx.x; i < m; i += 256) v[i] *= inv;
}

__global__ __launch_bounds__(256) void sn_finish_u_kernel(const float* __restrict__ part, int nsplit, int n,
                                                          float* __restrict__ u_new, float* __restrict__ aux) {
    __shared__ float sh[4];
    __shared__ float sc;
    float ss = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) {
        float b = 0.f;
        for (int s = 0; s < nsplit; ++s) b += part[(long)s * n + j];
        u_new[j] = b;
        ss += b * b;
    }
    const float t = block_sum_256(ss, sh);
    if (threadIdx.x == 0) {
        const float rb = sqrtf(t);
        sc = rb;
        aux[2] = rb;
        aux[0] = t / (rb + 1e-12f);      // sigma = b . (b/(|b|+eps))
    }
    __syncthreads();
    const float inv = 1.f / (sc + 1e-12f);
    for (int j = threadIdx.x; j < n; j += 256) u_new[j] *= inv;
}

__global__ void sn_scale_kernel(const float* __restrict__ W, const float* __restrict__ aux, long count,
                                float* __restrict__ wbar) {
    const float sigma = aux[0];
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) wbar[i] = W[i] / sigma;
}

extern "C" int ssc_sn_forward_any(const float* W, const float* u, int m, int n, float* v, float* u_new, float* wbar,
                                  float* aux, float* workspace, int64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int nsplit = (m + 127) / 128;
    if (nsplit > 64) nsplit = 64;
    if ((int64_t)nsplit * n * 4 > workspace_bytes) return -2;
    hipLaunchKernelGGL(sn_rowdot_kernel, dim3((m + 3) / 4), dim3(256), 0, st, W, u, m, n, v);
    hipLaunchKernelGGL(sn_normalize_v_kernel, dim3(1), dim3(256), 0, st, v, m, aux);
    hipLaunchKernelGGL(sn_coldot_partial_kernel, dim3((n + 63) / 64, nsplit), dim3(256), 0, st, W, v, m, n, nsplit,
                       workspace);
    hipLaunchKernelGGL(sn_finish_u_kernel, dim3(1), dim3(256), 0, st, workspace, nsplit, n, u_new, aux);
    long blocks = ((long)m * n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sn_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, st, W, aux, (long)m * n, wbar);
    return CHECK_LAUNCH();
}

__global__ void sn_dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long count,
                                      float* __restrict__ part) {
    __shared__ float sh[4];
    float acc = 0.f;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) acc += a[i] * b[i];
    const float t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// gb[j] = dsig * dsds * b_j / rb with dsig = -<G,W>/sigma^2
__global__ __launch_bounds__(256) void sn_gb_kernel(const float* __restrict__ part, int nparts,
                                                    const float* __restrict__ u_new, const float* __restrict__ aux,
                                                    int n, float* __restrict__ gb) {
    __shared__ float sh[4];
    __shared__ float sc;
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
    const float gw = block_sum_256(acc, sh);
    if (threadIdx.x == 0) sc = gw;
    __syncthreads();
    const float sigma = aux[0], rb = aux[2], eps = 1e-12f;
    const float dsig = -sc / (sigma * sigma);
    const float dsds = (rb * rb + 2.f * rb * eps) / ((rb + eps) * (rb + eps));
    for (int j = threadIdx.x; j < n; j += 256) gb[j] = dsig * dsds * (u_new[j] * (rb + eps)) / rb;
}

// scratch[i] = gv[i] on entry; ga[i] on exit
__global__ __launch_bounds__(256) void sn_ga_kernel(float* __restrict__ scratch, const float* __restrict__ v,
                                                    const float* __restrict__ aux, int m) {
    __shared__ float sh[4];
    __shared__ float sc;
    const float ra = aux[1], eps = 1e-12f;
    float dot = 0.f;
    for (int i = threadIdx.x; i < m; i += 256) dot += scratch[i] * v[i] * (ra + eps);
    const float t = block_sum_256(dot, sh);
    if (threadIdx.x == 0) sc = t;
    __syncthreads();
    const float gvdot = sc;
    for (int i = threadIdx.x; i < m; i += 256) {
        const float ai = v[i] * (ra + eps);
        scratch[i] = scratch[i] / (ra + eps) - gvdot * ai / (ra * (ra + eps) * (ra + eps));
    }
}

__global__ void sn_dw_kernel(const float* __restrict__ G, const float* __restrict__ v, const float* __restrict__ gb,
                             const float* __restrict__ ga, const float* __restrict__ u, const float* __restrict__ aux,
                             int m, int n, float* __restrict__ dW, int accumulate) {
    const float sigma = aux[0];
    const long count = (long)m * n, stride = (long)gridDim.x * blockDim.x;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const int i = (int)(k / n), j = (int)(k - (long)i * n);
        float d = G[k] / sigma + v[i] * gb[j] + ga[i] * u[j];
        if (accumulate) d += dW[k];
        dW[k] = d;
    }
}

// dW (+)= G/sigma + v^T gb + ga^T u; workspace >= (1024 + n) floats, scratch m floats
extern "C" int ssc_sn_backward_any(const float* W, const float* u, const float* v, const float* u_new, const float* aux,
                                   const float* G, int m, int n, float* dW, int accumulate, float* scratch,
                                   float* workspace, int64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int nparts = 1024;
    if ((int64_t)(nparts + n) * 4 > workspace_bytes) return -2;
    float* part = workspace;
    float* gb = workspace + nparts;
    hipLaunchKernelGGL(sn_dot_partial_kernel, dim3(nparts), dim3(256), 0, st, G, W, (long)m * n, part);
    hipLaunchKernelGGL(sn_gb_kernel, dim3(1), dim3(256), 0, st, part, nparts, u_new, aux, n, gb);
    hipLaunchKernelGGL(sn_rowdot_kernel, dim3((m + 3) / 4), dim3(256), 0, st, W, gb, m, n, scratch);
    hipLaunchKernelGGL(sn_ga_kernel, dim3(1), dim3(256), 0, st, scratch, v, aux, m);
    long blocks = ((long)m * n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sn_dw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, G, v, gb, scratch, u, aux, m, n, dW,
                       accumulate);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ Background_Colorization losses (bg_colorization_main.py:596-627)
// vanilla GAN terms on sigmoid(z): mode 0: -log(p + eps)  (real term of the discriminator loss / generator GAN loss),
// mode 1: -log(1 - p + eps) (fake term of the discriminator loss).  loss_acc += scale * sum;  dz = gscale * d/dz.
__global__ __launch_bounds__(256) void bg_gan_loss_kernel(const float* __restrict__ z, long n, int mode, float scale,
                                                           double* __restrict__ loss_acc, float* __restrict__ dz,
                                                           float gscale) {
    __shared__ float sh[4];
    const float eps = 1e-12f;
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float p = 1.f / (1.f + expf(-z[i]));
        const float dp = p * (1.f - p);
        if (mode == 0) {
            s += -logf(p + eps);
            if (dz != nullptr) dz[i] = gscale * (-dp / (p + eps));
        } else {
            s += -logf(1.f - p + eps);
            if (dz != nullptr) dz[i] = gscale * (dp / (1.f - p + eps));
        }
    }
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) atomicAdd(loss_acc, (double)scale * (double)t);
}

extern "C" int ssc_bg_gan_loss(const float* z, int64_t n, int mode, float scale, double* loss_acc, float* dz,
                               float gscale, void* stream) {
    long blocks = (n + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(bg_gan_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z, (long)n, mode,
                       scale, loss_acc, dz, gscale);
    return CHECK_LAUNCH();
}

// count[0] = number of labels != 0 (the pixels the L1 term averages over, :612-616)
__global__ __launch_bounds__(256) void count_nonzero_kernel(const int* __restrict__ labels, long n, float* __restrict__ part) {
    __shared__ float sh[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += labels[i] != 0 ? 1.f : 0.f;
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void count_fold_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[0] = t;
}

extern "C" int ssc_count_nonzero_i32(const int32_t* labels, int64_t n, float* count, float* workspace,
                                     int64_t workspace_bytes, void* stream) {
    const int blocks = 256;
    if (blocks * 4 > workspace_bytes) return -2;
    hipLaunchKernelGGL(count_nonzero_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, labels, (long)n, workspace);
    hipLaunchKernelGGL(count_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, blocks, count);
    return CHECK_LAUNCH();
}

// generator output: img = tanh(pre) [M,3].  loss_acc += l1w * mean_{label != 0} |t - o|;  dpre[M,4] =
// (l1 gradient + dgan[M,4]) * (1 - img^2), pad channel 0.
__global__ __launch_bounds__(256) void bg_output_grad_kernel(const float* __restrict__ img, const float* __restrict__ tgt,
                                                              const int* __restrict__ labels,
                                                              const float* __restrict__ count, float l1w,
                                                              const float* __restrict__ dgan,
                                                              double* __restrict__ loss_acc, float* __restrict__ dpre,
                                                              long M) {
    __shared__ float sh[4];
    const float denom = count[0] * 3.f;
    float s = 0.f;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long)gridDim.x * 256) {
        const bool sel = labels[r] != 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float o = img[r * 3 + c], t = tgt[r * 3 + c];
            float g = dgan != nullptr ? dgan[r * 4 + c] : 0.f;
            if (sel) {
                const float d = t - o;
                s += fabsf(d);
                g += l1w / denom * (d > 0.f ? -1.f : (d < 0.f ? 1.f : 0.f));
            }
            dpre[r * 4 + c] = g * (1.f - o * o);
        }
        dpre[r * 4 + 3] = 0.f;
    }
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) atomicAdd(loss_acc, (double)l1w * (double)t / (double)denom);
}

extern "C" int ssc_bg_output_grad(const float* img, const float* tgt, const int32_t* labels, const float* count,
                                  float l1w, const float* dgan, double* loss_acc, float* dpre, int64_t M, void* stream) {
    long blocks = (M + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(bg_output_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img, tgt, labels,
                       count, l1w, dgan, loss_acc, dpre, (long)M);
    return CHECK_LAUNCH();
}

// region-mask loss (:589-591): loss_acc += w * mean CE(softmax(logits[r, 0:K]), labels[r]);  dlogits[r, 0:ldg] (pad 0)
__global__ __launch_bounds__(256) void seg_ce_loss_kernel(const float* __restrict__ logits, int K,
                                                           const int* __restrict__ labels, long M, float w,
                                                           double* __restrict__ loss_acc, float* __restrict__ dlogits,
                                                           int ldg) {
    __shared__ float sh[4];
    float s = 0.f;
    const float inv = w / (float)M;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long)gridDim.x * 256) {
        float z[4], m = -INFINITY;
        for (int k = 0; k < K; ++k) { z[k] = logits[r * K + k]; m = fmaxf(m, z[k]); }
        float se = 0.f;
        for (int k = 0; k < K; ++k) { z[k] = expf(z[k] - m); se += z[k]; }
        const int y = labels[r];
        s += -(logits[r * K + y] - m - logf(se));
        for (int k = 0; k < ldg; ++k) dlogits[r * ldg + k] = k < K ? inv * (z[k] / se - (k == y ? 1.f : 0.f)) : 0.f;
    }
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) atomicAdd(loss_acc, (double)inv * (double)t);
}

extern "C" int ssc_seg_ce_loss(const float* logits, int K, const int32_t* labels, int64_t M, float w, double* loss_acc,
                               float* dlogits, int ldg, void* stream) {
    if (K > 4) return -1;
    long blocks = (M + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(seg_ce_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, K, labels,
                       (long)M, w, loss_acc, dlogits, ldg);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ flat-buffer helpers
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] += a * x[i];
}

extern "C" int ssc_axpy(float* y, const float* x, float a, int64_t n, void* stream) {
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, x, a, (long)n);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ small dense head (fully_connected, mru.py:52-92)
// y[n][j] = sum_k x[n][k] W[k][j] + b[j]; one wavefront per sample, J <= 64 outputs (class logits, J = 25).
// block per sample: thread = (output j = tid & 63, k-lane = tid >> 6), LDS reduce over the 4 k-lanes
__global__ __launch_bounds__(256) void fc_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ b, int K, int J,
                                                            float* __restrict__ y) {
    __shared__ float sh[256];
    const int n = blockIdx.x, j = threadIdx.x & 63, kl = threadIdx.x >> 6;
    float acc = 0.f;
    if (j < J) {
        // 8 independent partial sums: the loop is latency-bound (32 workgroups on the chip), the loads must overlap
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = kl;
        for (; k + 28 < K; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] = fmaf(x[(long)n * K + k + 4 * u], W[(long)(k + 4 * u) * J + j], a8[u]);
        }
        for (; k < K; k += 4) a8[0] = fmaf(x[(long)n * K + k], W[(long)k * J + j], a8[0]);
        acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (kl == 0 && j < J)
        y[(long)n * J + j] = sh[j] + sh[64 + j] + sh[128 + j] + sh[192 + j] + (b != nullptr ? b[j] : 0.f);
}

// dx[n][k] = sum_j dy[n][j] W[k][j];  dW[k][j] (+)= sum_n x[n][k] dy[n][j];  db[j] (+)= sum_n dy[n][j]
__global__ void fc_small_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ W, int N, int K, int J,
                                       float* __restrict__ dx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * K) return;
    const int n = (int)(i / K), k = (int)(i - (long)n * K);
    float acc = 0.f;
    for (int j = 0; j < J; ++j) acc += dy[(long)n * J + j] * W[(long)k * J + j];
    dx[i] = acc;
}

__global__ void fc_small_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy, int N, int K, int J,
                                       float* __restrict__ dW, float* __restrict__ db, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)(K + 1) * J) return;
    const int k = (int)(i / J), j = (int)(i - (long)k * J);
    float acc = 0.f;
    if (k < K) {
        for (int n = 0; n < N; ++n) acc += x[(long)n * K + k] * dy[(long)n * J + j];
        if (accumulate) acc += dW[i];
        dW[i] = acc;
    } else if (db != nullptr) {
        for (int n = 0; n < N; ++n) acc += dy[(long)n * J + j];
        if (accumulate) acc += db[j];
        db[j] = acc;
    }
}

extern "C" int ssc_fc_small_fwd(const float* x, const float* W, const float* b, int N, int K, int J, float* y,
                                void* stream) {
    if (J > 64) return -1;
    hipLaunchKernelGGL(fc_small_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, W, b, K, J, y);
    return CHECK_LAUNCH();
}

extern "C" int ssc_fc_small_bwd(const float* x, const float* W, const float* dy, int N, int K, int J, float* dx,
                                float* dW, float* db, int accumulate, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (dx != nullptr) {
        const long tot = (long)N * K;
        hipLaunchKernelGGL(fc_small_bwd_dx_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, dy, W, N, K, J,
                           dx);
    }
    if (dW != nullptr) {
        const long tot = (long)(K + 1) * J;
        hipLaunchKernelGGL(fc_small_bwd_dw_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, x, dy, N, K, J,
                           dW, db, accumulate);
    }
    return CHECK_LAUNCH();
}
