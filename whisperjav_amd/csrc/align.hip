// align.hip -- word-timestamp alignment after the teacher-forced decoder pass (gfx950).
//
// Input: the scaled cross-attention scores of the alignment heads, dumped by the decode cross-attention
// kernels as qk[row][sel][t][n_ctx] (fp32).  Restates whisper/timing.py find_alignment (which CTranslate2's
// Whisper.align re-implements; reference entry: word_timestamps=True through
// /root/reference/whisperjav/modules/faster_whisper_pro_asr.py:819):
//   softmax over the first num_frames/2 positions -> (w - mean_t) / std_t over the tokens axis (population std)
//   -> median filter of odd width along time, reflect padding -> mean over heads -> rows [n0, n_tok - 1)
//   -> DTW of the negated matrix with timing.py's strict-less tie-breaking -> (text index, time index) path.
// The DTW is an anti-diagonal wavefront (a cell depends on its three predecessors only, so the fp32 sums are
// the same numbers a row-by-row CPU loop produces: the path is bit-exact against the oracle for equal inputs).
#include "kernels.hpp"

namespace wj {

// in-place softmax over the first nf2[row] entries of every (row, sel, t < n_tok[row]) line
__global__ __launch_bounds__(256) void align_softmax_kernel(float* qk, const int32_t* n_tok, const int32_t* nf2, int nsel,
                                                            int tmax, int nctx) {
  const int t = blockIdx.x, sel = blockIdx.y, row = blockIdx.z;
  if (t >= n_tok[row]) return;
  const int n = nf2[row];
  float* x = qk + (((int64_t)row * nsel + sel) * tmax + t) * nctx;
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY;
  for (int j = tid; j < n; j += 256) mx = fmaxf(mx, x[j]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < n; j += 256) {
    const float e = expf(x[j] - mx);
    x[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
  for (int j = tid; j < n; j += 256) x[j] *= inv;
}

// (w - mean) / std over the tokens axis, per (row, sel, frame); population std, two passes
__global__ __launch_bounds__(256) void align_normalise_kernel(float* qk, const int32_t* n_tok, const int32_t* nf2, int nsel,
                                                              int tmax, int nctx) {
  const int sel = blockIdx.y, row = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= nf2[row]) return;
  const int n = n_tok[row];
  float* x = qk + ((int64_t)row * nsel + sel) * tmax * nctx + j;
  float mean = 0.f;
  for (int t = 0; t < n; ++t) mean += x[(int64_t)t * nctx];
  mean /= (float)n;
  float var = 0.f;
  for (int t = 0; t < n; ++t) {
    const float dlt = x[(int64_t)t * nctx] - mean;
    var = fmaf(dlt, dlt, var);
  }
  const float inv = 1.0f / sqrtf(var / (float)n);
  for (int t = 0; t < n; ++t) x[(int64_t)t * nctx] = (x[(int64_t)t * nctx] - mean) * inv;
}

// median filter along time (reflect padding) per head, then the mean over heads; writes matrix[row][t - n0][j]
// for n0 <= t < n_tok - 1
__global__ __launch_bounds__(256) void align_median_mean_kernel(const float* qk, float* matrix, const int32_t* n_tok,
                                                                const int32_t* nf2, int nsel, int tmax, int nctx, int n0,
                                                                int width) {
  const int t = blockIdx.y + n0, row = blockIdx.z;
  if (t >= n_tok[row] - 1) return;
  const int n = nf2[row];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int pad = width >> 1;
  float acc = 0.f;
  for (int sel = 0; sel < nsel; ++sel) {
    const float* x = qk + (((int64_t)row * nsel + sel) * tmax + t) * nctx;
    float v;
    if (n <= pad) {
      v = x[j];                        // timing.py: too short to pad -> returned unfiltered
    } else {
      float w[15];
      for (int k = 0; k < width; ++k) {
        int idx = j - pad + k;
        if (idx < 0) idx = -idx;                    // reflect without repeating the edge
        if (idx >= n) idx = 2 * (n - 1) - idx;
        w[k] = x[idx];
      }
      for (int a = 1; a < width; ++a) {             // insertion sort of <= 15 values
        const float key = w[a];
        int b = a - 1;
        while (b >= 0 && w[b] > key) { w[b + 1] = w[b]; --b; }
        w[b + 1] = key;
      }
      v = w[pad];
    }
    acc += v;
  }
  matrix[((int64_t)row * tmax + (t - n0)) * nctx + j] = acc / (float)nsel;
}

// DTW of -matrix, one workgroup per window.  cost lives in three rolling anti-diagonals in LDS, the trace
// (0 diagonal, 1 up, 2 left) in global memory; thread 0 walks it back.  Path entries are written in forward
// order to path_text / path_time [row][tmax + nctx], their count to path_len[row].
__global__ __launch_bounds__(256) void align_dtw_kernel(const float* matrix, int8_t* trace, const int32_t* n_tok,
                                                        const int32_t* nf2, int tmax, int nctx, int n0, int32_t* path_text,
                                                        int32_t* path_time, int32_t* path_len) {
  extern __shared__ float diag[];                   // [3][tmax + 1]
  const int row = blockIdx.x;
  const int N = n_tok[row] - 1 - n0, M = nf2[row];
  const int ld = nctx + 1;
  int8_t* tr = trace + (int64_t)row * (tmax + 1) * ld;
  const float* x = matrix + (int64_t)row * tmax * nctx;
  const int W = tmax + 1;
  if (N <= 0 || M <= 0) {
    if (threadIdx.x == 0) path_len[row] = 0;
    return;
  }
  // diagonal d holds cells (i, j = d - i); buffers rotate: d % 3
  for (int i = threadIdx.x; i < 3 * W; i += 256) diag[i] = INFINITY;
  __syncthreads();
  if (threadIdx.x == 0) diag[0] = 0.f;              // cost[0][0] on diagonal 0
  __syncthreads();
  for (int d = 1; d <= N + M; ++d) {
    float* cur = diag + (d % 3) * W;
    const float* p1 = diag + ((d + 2) % 3) * W;     // diagonal d - 1
    const float* p2 = diag + ((d + 1) % 3) * W;     // diagonal d - 2
    const int ilo = max(0, d - M), ihi = min(N, d);
    for (int i = ilo + threadIdx.x; i <= ihi; i += 256) {
      const int j = d - i;
      float v = INFINITY;
      if (i >= 1 && j >= 1) {
        const float c0 = p2[i - 1];                 // cost[i-1][j-1]
        const float c1 = p1[i - 1];                 // cost[i-1][j]
        const float c2 = p1[i];                     // cost[i][j-1]
        float c; int8_t tdir;
        if (c0 < c1 && c0 < c2) { c = c0; tdir = 0; }
        else if (c1 < c0 && c1 < c2) { c = c1; tdir = 1; }
        else { c = c2; tdir = 2; }
        v = -x[(int64_t)(i - 1) * nctx + (j - 1)] + c;
        tr[(int64_t)i * ld + j] = tdir;
      }
      cur[i] = v;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int i = N, j = M, n = 0;
    int32_t* pt = path_text + (int64_t)row * (tmax + nctx);
    int32_t* pf = path_time + (int64_t)row * (tmax + nctx);
    while (i > 0 || j > 0) {                        // backwards into the tail of the arrays, compacted below
      const int slot = tmax + nctx - 1 - n;
      pt[slot] = i - 1;
      pf[slot] = j - 1;
      ++n;
      const int tdir = (i == 0) ? 2 : (j == 0 ? 1 : tr[(int64_t)i * ld + j]);
      if (tdir == 0) { --i; --j; }
      else if (tdir == 1) --i;
      else --j;
    }
    const int off = tmax + nctx - n;
    for (int k = 0; k < n; ++k) { pt[k] = pt[off + k]; pf[k] = pf[off + k]; }
    path_len[row] = n;
  }
}

// probability of the token that follows: softmax over logits[:limit] gathered at tokens[row][pos + 1]
__global__ __launch_bounds__(256) void align_token_prob_kernel(const float* logits, int64_t ldl, int limit, const int32_t* tokens,
                                                               int64_t tok_stride, const int* pos_ptr, int n0, float* prob_out,
                                                               int tmax) {
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pos = *pos_ptr;
  if (pos < n0 || pos + 1 >= tmax) return;
  const float* x = logits + (int64_t)row * ldl;
  __shared__ float red[4];
  float mx = -INFINITY;
  for (int v = tid; v < limit; v += 256) mx = fmaxf(mx, x[v]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int v = tid; v < limit; v += 256) sum += expf(x[v] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  if (tid == 0) {
    const int tok = tokens[(int64_t)row * tok_stride + pos + 1];
    const float total = (red[0] + red[1]) + (red[2] + red[3]);
    prob_out[(int64_t)row * tmax + (pos - n0)] = (tok >= 0 && tok < limit) ? expf(x[tok] - mx) / total : 0.f;
  }
}

// full-sequence pass: logits row i of the chunk is global row row0 + i = (window, t); the probability of the token at
// t + 1 goes to prob_out[window][t - n0] for n0 <= t < n_tok[window] - 2 (the text-token predictions)
__global__ __launch_bounds__(256) void align_token_prob_seq_kernel(const float* logits, int64_t ldl, int limit,
                                                                   const int32_t* tokens, int64_t tok_stride, int row0, int Tp,
                                                                   int n0, const int32_t* n_tok, float* prob_out) {
  const int r = row0 + blockIdx.x, w = r / Tp, t = r % Tp;
  if (t < n0 || t >= n_tok[w] - 2) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = logits + (int64_t)blockIdx.x * ldl;
  __shared__ float red[4];
  float mx = -INFINITY;
  for (int v = tid; v < limit; v += 256) mx = fmaxf(mx, x[v]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int v = tid; v < limit; v += 256) sum += expf(x[v] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  if (tid == 0) {
    const int tok = tokens[(int64_t)w * tok_stride + t + 1];
    const float total = (red[0] + red[1]) + (red[2] + red[3]);
    prob_out[(int64_t)w * Tp + (t - n0)] = (tok >= 0 && tok < limit) ? expf(x[tok] - mx) / total : 0.f;
  }
}

int launch_align_token_prob_seq(const float* logits, int64_t ldl, int limit, const int32_t* tokens, int64_t tok_stride,
                                int row0, int rows, int Tp, int n0, const int32_t* n_tok, float* prob_out, hipStream_t s) {
  hipLaunchKernelGGL(align_token_prob_seq_kernel, dim3(rows), dim3(256), 0, s, logits, ldl, limit, tokens, tok_stride, row0, Tp,
                     n0, n_tok, prob_out);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

int launch_align_token_prob(const float* logits, int64_t ldl, int limit, const int32_t* tokens, int64_t tok_stride,
                            const int* pos_ptr, int n0, float* prob_out, int tmax, int R, hipStream_t s) {
  hipLaunchKernelGGL(align_token_prob_kernel, dim3(R), dim3(256), 0, s, logits, ldl, limit, tokens, tok_stride, pos_ptr, n0,
                     prob_out, tmax);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

int launch_align_post(float* qk, float* matrix, int8_t* trace, const int32_t* n_tok, const int32_t* nf2, int R, int nsel,
                      int tmax, int nctx, int n0, int width, int32_t* path_text, int32_t* path_time, int32_t* path_len,
                      hipStream_t s) {
  if (width < 1 || width > 15 || !(width & 1)) { set_error("align: median filter width %d must be odd and <= 15", width); return WJ_E_INVALID; }
  hipLaunchKernelGGL(align_softmax_kernel, dim3(tmax, nsel, R), dim3(256), 0, s, qk, n_tok, nf2, nsel, tmax, nctx);
  WJ_LAUNCH_CHECK();
  hipLaunchKernelGGL(align_normalise_kernel, dim3(ceil_div(nctx, 256), nsel, R), dim3(256), 0, s, qk, n_tok, nf2, nsel, tmax, nctx);
  WJ_LAUNCH_CHECK();
  hipLaunchKernelGGL(align_median_mean_kernel, dim3(ceil_div(nctx, 256), tmax, R), dim3(256), 0, s, qk, matrix, n_tok, nf2, nsel,
                     tmax, nctx, n0, width);
  WJ_LAUNCH_CHECK();
  hipLaunchKernelGGL(align_dtw_kernel, dim3(R), dim3(256), sizeof(float) * 3 * (tmax + 1), s, matrix, trace, n_tok, nf2, tmax, nctx,
                     n0, path_text, path_time, path_len);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

}  // namespace wj
