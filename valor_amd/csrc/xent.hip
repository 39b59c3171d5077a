// Fused softmax cross-entropy over the 30522-way vocabulary, forward + backward.
// Replaces F.cross_entropy on the gathered masked rows (model/pretrain.py:444,457,469,498;
// prediction scores from BERTPredictionHead modeling.py:245-254).
//   forward : lse[i] = logsumexp(logits[i, :V]);  loss[i] = lse[i] - logits[i, label[i]]
//   backward: dlogits[i, j] = (exp(logits[i,j] - lse[i]) - [j == label[i]]) * g,
//             g = (*gscale_dev) * gmul     (upstream scalar grad stays on the device: no host sync)
//             written IN PLACE over the logits (also zero-fills the ld padding).
// Label smoothing (LabelSmoothing of model/pretrain.py:46-61, the caption finetune loss when config.label_smoothing > 0, :839-840): the
// target is t[label] = 1 - eps, t[j] = eps / (V - 1) elsewhere and the row loss KL(t || softmax) = sum_j t_j (log t_j - logp_j)
//   = (1 - eps) log(1 - eps) + eps log(eps / (V - 1)) - (1 - eps) logp[label] - eps / (V - 1) * (sum_j logp_j - logp[label]),
//   logp_j = logits[j] - lse, sum_j logp_j = sum_j logits[j] - V lse;  dlogits[j] = (softmax_j - t_j) * g.   eps = 0 is the plain CE.
// One 256-thread workgroup per row; fp32 statistics.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void xent_fwd_kernel(const T* logits, const int64_t* labels, float* loss, float* lse,
                                                       int V, int64_t ld, float eps) {
    __shared__ float red[4], red2[4];
    const int64_t row = blockIdx.x;
    const T* x = logits + row * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int j = tid; j < V; j += 256) mx = fmaxf(mx, to_f32<T>(x[j]));
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f, zs = 0.f;
    for (int j = tid; j < V; j += 256) {
        const float z = to_f32<T>(x[j]);
        s += expf(z - mx);
        zs += z;
    }
    s = wave_sum(s);
    if (eps > 0.f) zs = wave_sum(zs);
    if (lane == 0) { red[wave] = s; red2[wave] = zs; }
    __syncthreads();
    if (tid == 0) {
        const float l = mx + logf(red[0] + red[1] + red[2] + red[3]);
        lse[row] = l;
        const int64_t lab = labels[row];
        const bool ok = lab >= 0 && lab < V;
        float v = ok ? l - to_f32<T>(x[lab]) : 0.f;            // -logp[label]
        if (eps > 0.f && ok) {
            const float u = eps / (float)(V - 1);
            const float sum_logp = (red2[0] + red2[1] + red2[2] + red2[3]) - (float)V * l;
            v = (1.0f - eps) * logf(1.0f - eps) + eps * logf(u) + (1.0f - eps) * v - u * (sum_logp + v);
        }
        loss[row] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void xent_bwd_kernel(T* logits, const int64_t* labels, const float* lse,
                                                       const float* gscale_dev, float gmul, int V, int64_t ld, float eps) {
    const int64_t row = blockIdx.x;
    T* x = logits + row * ld;
    const float l = lse[row];
    const int64_t lab = labels[row];
    const float g = (gscale_dev ? *gscale_dev : 1.0f) * gmul;
    const float t_other = eps > 0.f ? eps / (float)(V - 1) : 0.f, t_label = 1.0f - eps;
    for (int j = threadIdx.x; j < (int)ld; j += 256) {
        float d = 0.f;
        if (j < V) {
            d = expf(to_f32<T>(x[j]) - l);
            d -= j == lab ? t_label : t_other;
            d *= g;
        }
        x[j] = from_f32<T>(d);
    }
}

extern "C" int valor_xent_smooth_fwd(void* stream, int dtype, const void* logits, const int64_t* labels, float* loss_rows,
                                     float* lse, int64_t rows, int V, int64_t ld, float smoothing) {
    if (rows <= 0) return VALOR_OK;
    if (!logits || !labels || !loss_rows || !lse || V <= 0 || !(smoothing >= 0.f && smoothing < 1.f) || (smoothing > 0.f && V <= 1)) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VALOR_DT_BF16)
        hipLaunchKernelGGL((xent_fwd_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, (const bf16_t*)logits, labels, loss_rows, lse, V, ld, smoothing);
    else if (dtype == VALOR_DT_F32)
        hipLaunchKernelGGL((xent_fwd_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, labels, loss_rows, lse, V, ld, smoothing);
    else return VALOR_ERR_ARG;
    return valor_launch_status();
}
extern "C" int valor_xent_fwd(void* stream, int dtype, const void* logits, const int64_t* labels, float* loss_rows,
                              float* lse, int64_t rows, int V, int64_t ld) {
    return valor_xent_smooth_fwd(stream, dtype, logits, labels, loss_rows, lse, rows, V, ld, 0.f);
}

extern "C" int valor_xent_smooth_bwd(void* stream, int dtype, void* logits_inout, const int64_t* labels, const float* lse,
                                     const float* gscale_dev, float gmul, int64_t rows, int V, int64_t ld, float smoothing) {
    if (rows <= 0) return VALOR_OK;
    if (!logits_inout || !labels || !lse || V <= 0 || !(smoothing >= 0.f && smoothing < 1.f) || (smoothing > 0.f && V <= 1)) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VALOR_DT_BF16)
        hipLaunchKernelGGL((xent_bwd_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, (bf16_t*)logits_inout, labels, lse, gscale_dev, gmul, V, ld, smoothing);
    else if (dtype == VALOR_DT_F32)
        hipLaunchKernelGGL((xent_bwd_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, (float*)logits_inout, labels, lse, gscale_dev, gmul, V, ld, smoothing);
    else return VALOR_ERR_ARG;
    return valor_launch_status();
}
extern "C" int valor_xent_bwd(void* stream, int dtype, void* logits_inout, const int64_t* labels, const float* lse,
                              const float* gscale_dev, float gmul, int64_t rows, int V, int64_t ld) {
    return valor_xent_smooth_bwd(stream, dtype, logits_inout, labels, lse, gscale_dev, gmul, rows, V, ld, 0.f);
}
