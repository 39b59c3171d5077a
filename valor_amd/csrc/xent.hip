// Fused softmax cross-entropy over the 30522-way vocabulary, forward + backward.
// Replaces F.cross_entropy on the gathered masked rows (model/pretrain.py:444,457,469,498;
// prediction scores from BERTPredictionHead modeling.py:245-254).
//   forward : lse[i] = logsumexp(logits[i, :V]);  loss[i] = lse[i] - logits[i, label[i]]
//   backward: dlogits[i, j] = (exp(logits[i,j] - lse[i]) - [j == label[i]]) * g,
//             g = (*gscale_dev) * gmul     (upstream scalar grad stays on the device: no host sync)
//             written IN PLACE over the logits (also zero-fills the ld padding).
// One 256-thread workgroup per row; fp32 statistics.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void xent_fwd_kernel(const T* logits, const int64_t* labels, float* loss, float* lse,
                                                       int V, int64_t ld) {
    __shared__ float red[4];
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
    float s = 0.f;
    for (int j = tid; j < V; j += 256) s += expf(to_f32<T>(x[j]) - mx);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        const float l = mx + logf(red[0] + red[1] + red[2] + red[3]);
        lse[row] = l;
        const int64_t lab = labels[row];
        loss[row] = (lab >= 0 && lab < V) ? l - to_f32<T>(x[lab]) : 0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void xent_bwd_kernel(T* logits, const int64_t* labels, const float* lse,
                                                       const float* gscale_dev, float gmul, int V, int64_t ld) {
    const int64_t row = blockIdx.x;
    T* x = logits + row * ld;
    const float l = lse[row];
    const int64_t lab = labels[row];
    const float g = (gscale_dev ? *gscale_dev : 1.0f) * gmul;
    for (int j = threadIdx.x; j < (int)ld; j += 256) {
        float d = 0.f;
        if (j < V) {
            d = expf(to_f32<T>(x[j]) - l);
            if (j == lab) d -= 1.0f;
            d *= g;
        }
        x[j] = from_f32<T>(d);
    }
}

extern "C" int valor_xent_fwd(void* stream, int dtype, const void* logits, const int64_t* labels, float* loss_rows,
                              float* lse, int64_t rows, int V, int64_t ld) {
    if (rows <= 0) return VALOR_OK;
    if (!logits || !labels || !loss_rows || !lse || V <= 0) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VALOR_DT_BF16)
        hipLaunchKernelGGL((xent_fwd_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, (const bf16_t*)logits, labels, loss_rows, lse, V, ld);
    else if (dtype == VALOR_DT_F32)
        hipLaunchKernelGGL((xent_fwd_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, labels, loss_rows, lse, V, ld);
    else return VALOR_ERR_ARG;
    return valor_launch_status();
}

extern "C" int valor_xent_bwd(void* stream, int dtype, void* logits_inout, const int64_t* labels, const float* lse,
                              const float* gscale_dev, float gmul, int64_t rows, int V, int64_t ld) {
    if (rows <= 0) return VALOR_OK;
    if (!logits_inout || !labels || !lse || V <= 0) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VALOR_DT_BF16)
        hipLaunchKernelGGL((xent_bwd_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, (bf16_t*)logits_inout, labels, lse, gscale_dev, gmul, V, ld);
    else if (dtype == VALOR_DT_F32)
        hipLaunchKernelGGL((xent_bwd_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, (float*)logits_inout, labels, lse, gscale_dev, gmul, V, ld);
    else return VALOR_ERR_ARG;
    return valor_launch_status();
}
