// detect.hip -- base->novel similarity transfer (fine-tune / eval predictor) and the detection post-processing.
//   a14  WSROIHead.get_similarity_matrices  modeling/roi_heads/roi_heads.py:245-336  ('lingual' + 'visual', "Sum")
//        SupervisedDetectorOutputs*.forward transfer   modeling/roi_heads/fast_rcnn.py:401-423, 504-523
//   a15  SupervisedDetectorOutputsBase.inference -> detectron2 fast_rcnn_inference (fast_rcnn.py:455-468)
#include "common.h"

#define DET_MAXC 96

// out[i][j] = sum_k a[i][k] * b[j][k]   (tiny: label-embedding similarity 5x300 . 15x300^T, fast_rcnn.py:376-382)
__global__ void small_matmul_nt_kernel(const float* __restrict__ a, const int* __restrict__ arow, int M, const float* __restrict__ b,
                                       const int* __restrict__ brow, int N, int Kd, int lda, float* __restrict__ out) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * N) return;
  int i = idx / N, j = idx - i * N;
  const float* pa = a + (size_t)(arow ? arow[i] : i) * lda;
  const float* pb = b + (size_t)(brow ? brow[j] : j) * lda;
  float s = 0.f;
  for (int k = 0; k < Kd; ++k) s += pa[k] * pb[k];
  out[idx] = s;
}
extern "C" int unit_embedding_similarity(const float* emb, int ld, int dim, const int* novel_rows, int n_novel, const int* base_rows,
                                         int n_base, float* out, void* stream) {
  if (n_novel * n_base == 0) return UNIT_OK;
  small_matmul_nt_kernel<<<cdiv(n_novel * n_base, 128), 128, 0, (hipStream_t)stream>>>(emb, novel_rows, n_novel, emb, base_rows, n_base, dim, ld, out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// sim[r][j][b] for j in novel, b in base (roi_heads.py:250-257, 269-272, 316-322):
//   probs = mean_k oicr_k(x)           (K+1 logits)           vis = softmax(probs)[base] ; vis /= max(sum, 1e-9) ; vis[vis < thr] = 0
//   s = 0.5 * softmax(lingual[j,:]) + 0.5 * vis              s /= max(sum_b s, 1e-9)
__global__ void similarity_kernel(const float* __restrict__ lin, int ld, int col0, int n_oicr, int ncls, const int* __restrict__ base,
                                  int n_base, const float* __restrict__ lingual, int n_novel, float thr, int use_lingual,
                                  int use_visual, float* __restrict__ sim, int R) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float vis[DET_MAXC];
  if (use_visual) {
    const float* x = lin + (size_t)r * ld + col0;
    float mx = -INFINITY;
    for (int c = 0; c < ncls; ++c) {
      float s = 0.f;
      for (int k = 0; k < n_oicr; ++k) s += x[k * ncls + c];
      mx = fmaxf(mx, s / (float)n_oicr);
    }
    float se = 0.f;
    for (int c = 0; c < ncls; ++c) {
      float s = 0.f;
      for (int k = 0; k < n_oicr; ++k) s += x[k * ncls + c];
      se += expf(s / (float)n_oicr - mx);
    }
    float tot = 0.f;
    for (int b = 0; b < n_base; ++b) {
      int c = base[b];
      float s = 0.f;
      for (int k = 0; k < n_oicr; ++k) s += x[k * ncls + c];
      vis[b] = expf(s / (float)n_oicr - mx) / se;
      tot += vis[b];
    }
    tot = fmaxf(tot, 1e-9f);
    for (int b = 0; b < n_base; ++b) { float v = vis[b] / tot; vis[b] = v < thr ? 0.f : v; }
  }
  float nterms = (float)(use_lingual + use_visual);
  float wgt = nterms > 0.f ? 1.0f / nterms : 0.f;
  for (int j = 0; j < n_novel; ++j) {
    float lmx = -INFINITY, lse = 0.f;
    if (use_lingual) {
      for (int b = 0; b < n_base; ++b) lmx = fmaxf(lmx, lingual[j * n_base + b]);
      for (int b = 0; b < n_base; ++b) lse += expf(lingual[j * n_base + b] - lmx);
    }
    float tot = 0.f;
    float* o = sim + ((size_t)r * n_novel + j) * n_base;
    for (int b = 0; b < n_base; ++b) {
      float s = 0.f;
      if (use_lingual) s = s + wgt * (expf(lingual[j * n_base + b] - lmx) / lse);
      if (use_visual) s = s + wgt * vis[b];
      o[b] = s; tot += s;
    }
    tot = fmaxf(tot, 1e-9f);
    for (int b = 0; b < n_base; ++b) o[b] = nterms > 0.f ? o[b] / tot : 0.f;
  }
}
extern "C" int unit_similarity(const float* lin_weak, int ld, int col0, int n_oicr, int ncls, const int* base_dev, int n_base,
                               const float* lingual, int n_novel, float visual_threshold, int use_lingual, int use_visual,
                               float* sim, int R, void* stream) {
  UNIT_CHECK_ARG(n_base <= DET_MAXC, "similarity: more than 96 base classes");
  if (R == 0) return UNIT_OK;
  similarity_kernel<<<cdiv(R, 64), 64, 0, (hipStream_t)stream>>>(lin_weak, ld, col0, n_oicr, ncls, base_dev, n_base, lingual, n_novel,
                                                                visual_threshold, use_lingual, use_visual, sim, R);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// fast_rcnn.py:401-423 / 504-523 + :425-426 (+ :525-528 fine-tune heads):
//   scores[r,c] = delta[r,c] + [c novel] sum_b sim_cls[r,j,b] * delta[r,base_b]  + mean_k oicr_k  (+ ft)
//   bbox[r,c,:] = (c base) delta ; (c novel) sum_b sim_bbox[r,j,b] * delta[r,base_b,:] ; (else) 0   (+ 0 weak) (+ ft)
__global__ void transfer_kernel(const float* __restrict__ lin, int ld, int ccol0, int bcol0, int K, const float* __restrict__ weak, int ldw,
                                int wcol0, int n_oicr, const float* __restrict__ ft, int ldf, int fccol0, int fbcol0,
                                const float* __restrict__ sim_cls, const float* __restrict__ sim_bbox, const int* __restrict__ base, int n_base,
                                const int* __restrict__ novel, int n_novel, const int8_t* __restrict__ role /* K: 0 none,1 base,2 novel */,
                                const int* __restrict__ slot /* K: index into base/novel list */, float* __restrict__ scores, int lds_,
                                float* __restrict__ bbox, int ldb, int R) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int ncls = K + 1;
  if (idx >= R * ncls) return;
  int r = idx / ncls, c = idx - r * ncls;
  const float* x = lin + (size_t)r * ld;
  float s = x[ccol0 + c];
  if (c < K && sim_cls && role[c] == 2) {
    const float* sm = sim_cls + ((size_t)r * n_novel + slot[c]) * n_base;
    float t = 0.f;
    for (int b = 0; b < n_base; ++b) t += sm[b] * x[ccol0 + base[b]];
    s = s + t;
  }
  if (weak) {
    float w = 0.f;
    for (int k = 0; k < n_oicr; ++k) w += weak[(size_t)r * ldw + wcol0 + k * ncls + c];
    s = s + w / (float)n_oicr;
  }
  if (ft) s = s + ft[(size_t)r * ldf + fccol0 + c];
  scores[(size_t)r * lds_ + c] = s;
  if (c >= K) return;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  if (!sim_bbox || role[c] == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = x[bcol0 + 4 * c + j];
  } else if (role[c] == 2) {
    const float* sm = sim_bbox + ((size_t)r * n_novel + slot[c]) * n_base;
    for (int b = 0; b < n_base; ++b) {
      float w = sm[b];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] += w * x[bcol0 + 4 * base[b] + j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) bbox[(size_t)r * ldb + 4 * c + j] = o[j] + (ft ? ft[(size_t)r * ldf + fbcol0 + 4 * c + j] : 0.f);
}
extern "C" int unit_transfer_predictions(const float* lin, int ld, int ccol0, int bcol0, int K, const float* weak, int ldw, int wcol0,
                                         int n_oicr, const float* ft, int ldf, int fccol0, int fbcol0, const float* sim_cls,
                                         const float* sim_bbox, const int* base_dev, int n_base, const int* novel_dev, int n_novel,
                                         const int8_t* role_dev, const int* slot_dev, float* scores, int lds, float* bbox, int ldb,
                                         int R, void* stream) {
  if (R == 0) return UNIT_OK;
  transfer_kernel<<<cdiv(R * (K + 1), 256), 256, 0, (hipStream_t)stream>>>(lin, ld, ccol0, bcol0, K, weak, ldw, wcol0, n_oicr, ft, ldf, fccol0,
                                                                       fbcol0, sim_cls, sim_bbox, base_dev, n_base, novel_dev, n_novel,
                                                                       role_dev, slot_dev, scores, lds, bbox, ldb, R);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------
// Backward of the transfer (fine-tune configurations whose box head trains: COCO-RCNN-50-C4-split1-segm-ft.yaml). The reference
// computes the similarity WITHOUT no_grad in training (roi_heads.py:852 -> :245-336), so d(loss) flows
//   scores / bbox -> [delta heads' outputs (frozen weights, but their input is the trainable box head's feature), similarity]
//   dlin_cls[r,c]    = dsc[r,c] + [c base] sum_j sim_cls[r,j,slot c] dsc[r,novel_j]
//   dlin_bbox[r,c,:] = [c base] (dbb[r,c,:] + sum_j sim_bbox[r,j,slot c] dbb[r,novel_j,:])       (novel / other rows of the delta
//                      head are overwritten by the transfer: no gradient)
//   dsim[r,j,b]      = dsc[r,novel_j] lin_cls[r,base_b] + sum_k dbb[r,novel_j,k] lin_bbox[r,base_b,k]   (cls and bbox use the
//                      same matrix when their term lists are equal; `dsim` is WRITTEN here, the mask head adds to it afterwards)
// dy = d(loss)/d[scores | bbox] in the ft heads' column layout (dccol0 / dbcol0); dlin has the delta heads' layout.
// ---------------------------------------------------------------------------------------------------
template <typename TD>
__global__ void transfer_bwd_kernel(const TD* __restrict__ dy, int ldd, int dccol0, int dbcol0, const float* __restrict__ lin, int ld,
                                    int ccol0, int bcol0, int K, const float* __restrict__ sim_cls, const float* __restrict__ sim_bbox,
                                    const int* __restrict__ base, int n_base, const int* __restrict__ novel, int n_novel,
                                    const int8_t* __restrict__ role, const int* __restrict__ slot, TD* __restrict__ dlin, int ldl,
                                    float* __restrict__ dsim, int R) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int ncls = K + 1;
  int n1 = R * ncls, n2 = R * n_novel * n_base;
  if (idx < n1) {
    int r = idx / ncls, c = idx - r * ncls;
    const TD* d = dy + (size_t)r * ldd;
    float g = (float)d[dccol0 + c];
    float gb[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < K && role[c] == 1) {
      int sb = slot[c];
#pragma unroll
      for (int k = 0; k < 4; ++k) gb[k] = (float)d[dbcol0 + 4 * c + k];
      for (int j = 0; j < n_novel; ++j) {
        int cn = novel[j];
        size_t so = ((size_t)r * n_novel + j) * n_base + sb;
        g += sim_cls[so] * (float)d[dccol0 + cn];
        float wb = sim_bbox[so];
#pragma unroll
        for (int k = 0; k < 4; ++k) gb[k] += wb * (float)d[dbcol0 + 4 * cn + k];
      }
    }
    TD* o = dlin + (size_t)r * ldl;
    o[ccol0 + c] = (TD)g;
    if (c < K) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[bcol0 + 4 * c + k] = (TD)gb[k];
    }
  } else if (idx < n1 + n2) {
    int t = idx - n1;
    int r = t / (n_novel * n_base); int rem = t - r * n_novel * n_base; int j = rem / n_base, b = rem - j * n_base;
    const TD* d = dy + (size_t)r * ldd;
    const float* x = lin + (size_t)r * ld;
    int cn = novel[j], cb = base[b];
    float v = (float)d[dccol0 + cn] * x[ccol0 + cb];
#pragma unroll
    for (int k = 0; k < 4; ++k) v += (float)d[dbcol0 + 4 * cn + k] * x[bcol0 + 4 * cb + k];
    dsim[t] = v;
  }
}
extern "C" int unit_transfer_predictions_bwd(const void* dy, int dy_dtype, int ldd, int dccol0, int dbcol0, const float* lin, int ld,
                                             int ccol0, int bcol0, int K, const float* sim_cls, const float* sim_bbox, const int* base_dev,
                                             int n_base, const int* novel_dev, int n_novel, const int8_t* role_dev, const int* slot_dev,
                                             void* dlin, int ldl, float* dsim, int R, void* stream) {
  if (R == 0) return UNIT_OK;
  long n = (long)R * (K + 1) + (long)R * n_novel * n_base;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(dlin, 0, (size_t)R * ldl * (dy_dtype == UNIT_BF16 ? 2 : 4), st);
  if (dy_dtype == UNIT_BF16)
    transfer_bwd_kernel<bf16_t><<<cdiv(n, 256), 256, 0, st>>>((const bf16_t*)dy, ldd, dccol0, dbcol0, lin, ld, ccol0, bcol0, K, sim_cls, sim_bbox,
                                                            base_dev, n_base, novel_dev, n_novel, role_dev, slot_dev, (bf16_t*)dlin, ldl, dsim, R);
  else
    transfer_bwd_kernel<float><<<cdiv(n, 256), 256, 0, st>>>((const float*)dy, ldd, dccol0, dbcol0, lin, ld, ccol0, bcol0, K, sim_cls, sim_bbox,
                                                          base_dev, n_base, novel_dev, n_novel, role_dev, slot_dev, (float*)dlin, ldl, dsim, R);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// Backward of similarity_kernel: dsim [R][n_novel][n_base] -> d(mean OICR logits) -> the n_oicr logit groups of `dlin` (each gets
// 1 / n_oicr of it; the predictors' weights are frozen in every fine-tune yaml, the gradient continues into their input).
//   S = u / T, u[j,b] = w L[j,b] + w v_b, T_j = max(sum_b u, 1e-9);  v_b = v'_b (>= thr) | 0,  v' = q[base] / max(sum q[base], 1e-9),
//   q = softmax(p).   (the in-place zeroing of roi_heads.py:257 passes no gradient to the zeroed entries)
template <typename TD>
__global__ void similarity_bwd_kernel(const float* __restrict__ lin, int ld, int col0, int n_oicr, int ncls, const int* __restrict__ base,
                                      int n_base, const float* __restrict__ lingual, int n_novel, float thr, int use_lingual,
                                      int use_visual, const float* __restrict__ dsim, TD* __restrict__ dlin, int ldl, int dcol0, int R) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R || !use_visual) return;
  float vis[DET_MAXC], vraw[DET_MAXC], dv[DET_MAXC];
  const float* x = lin + (size_t)r * ld + col0;
  float mx = -INFINITY;
  for (int c = 0; c < ncls; ++c) {
    float s = 0.f;
    for (int k = 0; k < n_oicr; ++k) s += x[k * ncls + c];
    mx = fmaxf(mx, s / (float)n_oicr);
  }
  float se = 0.f;
  for (int c = 0; c < ncls; ++c) {
    float s = 0.f;
    for (int k = 0; k < n_oicr; ++k) s += x[k * ncls + c];
    se += expf(s / (float)n_oicr - mx);
  }
  float tot = 0.f;
  for (int b = 0; b < n_base; ++b) {
    int c = base[b];
    float s = 0.f;
    for (int k = 0; k < n_oicr; ++k) s += x[k * ncls + c];
    vraw[b] = expf(s / (float)n_oicr - mx) / se;          // q[base_b]
    tot += vraw[b];
  }
  float totc = fmaxf(tot, 1e-9f);
  for (int b = 0; b < n_base; ++b) { float v = vraw[b] / totc; vis[b] = v < thr ? 0.f : v; dv[b] = 0.f; }
  float nterms = (float)(use_lingual + use_visual);
  float wgt = 1.0f / nterms;
  for (int j = 0; j < n_novel; ++j) {
    float lmx = -INFINITY, lse = 0.f;
    if (use_lingual) {
      for (int b = 0; b < n_base; ++b) lmx = fmaxf(lmx, lingual[j * n_base + b]);
      for (int b = 0; b < n_base; ++b) lse += expf(lingual[j * n_base + b] - lmx);
    }
    float T = 0.f;
    for (int b = 0; b < n_base; ++b) {
      float u = use_lingual ? wgt * (expf(lingual[j * n_base + b] - lmx) / lse) : 0.f;
      T += u + wgt * vis[b];
    }
    float Tc = fmaxf(T, 1e-9f);
    const float* g = dsim + ((size_t)r * n_novel + j) * n_base;
    float dot = 0.f;                                      // sum_b G[j,b] S[j,b]
    if (T >= 1e-9f) {
      for (int b = 0; b < n_base; ++b) {
        float u = (use_lingual ? wgt * (expf(lingual[j * n_base + b] - lmx) / lse) : 0.f) + wgt * vis[b];
        dot += g[b] * (u / Tc);
      }
    }
    for (int b = 0; b < n_base; ++b) dv[b] += wgt * (g[b] - dot) / Tc;       // d u[j,b] = (G - sum G S) / T  (T clamped: dot = 0)
  }
  // v_b = v'_b (kept) -> v'_b = w_b / tot
  float dot2 = 0.f;
  for (int b = 0; b < n_base; ++b) { if (vis[b] == 0.f) dv[b] = 0.f; }
  if (tot >= 1e-9f) for (int b = 0; b < n_base; ++b) dot2 += dv[b] * (vraw[b] / totc);
  // dq[c] for base columns, then dp = q (dq - sum q dq)
  float sq = 0.f;
  for (int b = 0; b < n_base; ++b) { dv[b] = (dv[b] - dot2) / totc; sq += vraw[b] * dv[b]; }     // dv now = dq[base_b]; sq = sum_c q_c dq_c
  TD* o = dlin + (size_t)r * ldl + dcol0;
  for (int c = 0; c < ncls; ++c) {
    float s = 0.f;
    for (int k = 0; k < n_oicr; ++k) s += x[k * ncls + c];
    float q = expf(s / (float)n_oicr - mx) / se;
    float dq = 0.f;
    for (int b = 0; b < n_base; ++b) if (base[b] == c) dq = dv[b];
    float dp = q * (dq - sq) / (float)n_oicr;
    for (int k = 0; k < n_oicr; ++k) o[k * ncls + c] = (TD)dp;
  }
}
extern "C" int unit_similarity_bwd(const float* lin_weak, int ld, int col0, int n_oicr, int ncls, const int* base_dev, int n_base,
                                   const float* lingual, int n_novel, float visual_threshold, int use_lingual, int use_visual,
                                   const float* dsim, void* dlin, int dlin_dtype, int ldl, int dcol0, int R, void* stream) {
  UNIT_CHECK_ARG(n_base <= DET_MAXC, "similarity_bwd: more than 96 base classes");
  if (R == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(dlin, 0, (size_t)R * ldl * (dlin_dtype == UNIT_BF16 ? 2 : 4), st);
  if (dlin_dtype == UNIT_BF16)
    similarity_bwd_kernel<bf16_t><<<cdiv(R, 64), 64, 0, st>>>(lin_weak, ld, col0, n_oicr, ncls, base_dev, n_base, lingual, n_novel, visual_threshold,
                                                            use_lingual, use_visual, dsim, (bf16_t*)dlin, ldl, dcol0, R);
  else
    similarity_bwd_kernel<float><<<cdiv(R, 64), 64, 0, st>>>(lin_weak, ld, col0, n_oicr, ncls, base_dev, n_base, lingual, n_novel, visual_threshold,
                                                           use_lingual, use_visual, dsim, (float*)dlin, ldl, dcol0, R);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// row softmax (predict_probs: F.softmax(scores, dim=-1))
__global__ void softmax_rows_kernel(const float* __restrict__ x, int ld, int ncls, float* __restrict__ y, int ldy, int R) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* p = x + (size_t)r * ld;
  float mx = -INFINITY;
  for (int c = 0; c < ncls; ++c) mx = fmaxf(mx, p[c]);
  float se = 0.f;
  for (int c = 0; c < ncls; ++c) se += expf(p[c] - mx);
  for (int c = 0; c < ncls; ++c) y[(size_t)r * ldy + c] = expf(p[c] - mx) / se;
}
extern "C" int unit_softmax_rows(const float* x, int ld, int ncls, float* y, int ldy, int R, void* stream) {
  if (R == 0) return UNIT_OK;
  softmax_rows_kernel<<<cdiv(R, 128), 128, 0, (hipStream_t)stream>>>(x, ld, ncls, y, ldy, R);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// fast_rcnn_inference_single_image, first half: decode (weights w), clip, score > thresh -> candidates in (roi, class)
// row-major order, compacted. One workgroup per image. Also returns max coordinate over the candidates (batched_nms offset).
__device__ __forceinline__ int2 det_scan2(int2 v, int2* total, int2* lds) {
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int2 inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int ax = __shfl_up(inc.x, o, 64), ay = __shfl_up(inc.y, o, 64);
    if (lane >= o) { inc.x += ax; inc.y += ay; }
  }
  if (lane == 63) lds[wid] = inc;
  __syncthreads();
  if (threadIdx.x == 0) { int2 run = {0, 0}; for (int w = 0; w < nw; ++w) { int2 t = lds[w]; lds[w] = run; run.x += t.x; run.y += t.y; } lds[16] = run; }
  __syncthreads();
  int2 base = lds[wid];
  *total = lds[16];
  int2 ex = {base.x + inc.x - v.x, base.y + inc.y - v.y};
  __syncthreads();
  return ex;
}

__global__ void det_select_kernel(const float* __restrict__ probs, int ldp, const float* __restrict__ deltas, int ldd,
                                  const float* __restrict__ props, const int* __restrict__ pcount, int Rcap, int K, f32x4 w,
                                  float clampv, const float* __restrict__ image_hw, float thresh, int cap, float* __restrict__ cboxes,
                                  float* __restrict__ cscores, int* __restrict__ cclass, int* __restrict__ croi, int* __restrict__ ccount,
                                  float* __restrict__ cmax) {
  __shared__ int2 lds[17];
  __shared__ float smax[16];
  int b = blockIdx.x;
  int R = min(pcount ? pcount[b] : Rcap, Rcap);
  float imh = image_hw[2 * b], imw = image_hw[2 * b + 1];
  int total = R * K;
  int chunk = (total + blockDim.x - 1) / blockDim.x;
  int i0 = threadIdx.x * chunk, i1 = min(total, i0 + chunk);
  int2 c = {0, 0};
  float mx = -INFINITY;
  for (int pass = 0; pass < 2; ++pass) {
    int2 ex = {0, 0}, tot = {0, 0};
    if (pass == 1) ex = det_scan2(c, &tot, lds);
    for (int i = i0; i < i1; ++i) {
      int r = i / K, k = i - r * K;
      size_t row = (size_t)b * Rcap + r;
      float sc = probs[row * ldp + k];
      const float* dp = deltas + row * ldd + 4 * k;
      const float* pb = props + row * 4;
      f32x4 bx;
      {
        float bw = pb[2] - pb[0], bh = pb[3] - pb[1];
        float cx = pb[0] + 0.5f * bw, cy = pb[1] + 0.5f * bh;
        float dx = dp[0] / w[0], dy = dp[1] / w[1];
        float dw = fminf(dp[2] / w[2], clampv), dh = fminf(dp[3] / w[3], clampv);
        float pcx = dx * bw + cx, pcy = dy * bh + cy;
        float pw = expf(dw) * bw, ph = expf(dh) * bh;
        bx = f32x4{pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph};
      }
      bool fin = isfinite(bx[0]) && isfinite(bx[1]) && isfinite(bx[2]) && isfinite(bx[3]) && isfinite(sc);
      bx[0] = fminf(fmaxf(bx[0], 0.f), imw); bx[1] = fminf(fmaxf(bx[1], 0.f), imh);
      bx[2] = fminf(fmaxf(bx[2], 0.f), imw); bx[3] = fminf(fmaxf(bx[3], 0.f), imh);
      bool keep = fin && sc > thresh;
      if (pass == 0) { c.x += keep ? 1 : 0; }
      else if (keep) {
        if (ex.x < cap) {
          size_t o = (size_t)b * cap + ex.x;
          *reinterpret_cast<f32x4*>(cboxes + 4 * o) = bx;
          cscores[o] = sc; cclass[o] = k; croi[o] = r;
          mx = fmaxf(mx, fmaxf(fmaxf(bx[0], bx[1]), fmaxf(bx[2], bx[3])));
        }
        ex.x++;
      }
    }
    if (pass == 1 && threadIdx.x == 0) ccount[b] = min(tot.x, cap);
  }
  mx = wave_reduce_max(mx);
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) { float m = -INFINITY; for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) m = fmaxf(m, smax[w2]); cmax[b] = m; }
}
extern "C" int unit_detection_candidates(const float* probs, int ldp, const float* deltas, int ldd, const float* props, const int* pcount,
                                         int B, int Rcap, int K, const float* weights4, float scale_clamp, const float* image_hw_dev,
                                         float score_thresh, int cap, float* cand_boxes, float* cand_scores, int* cand_class, int* cand_roi,
                                         int* cand_count, float* cand_max, void* stream) {
  if (B == 0) return UNIT_OK;
  f32x4 w = {weights4[0], weights4[1], weights4[2], weights4[3]};
  det_select_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(probs, ldp, deltas, ldd, props, pcount, Rcap, K, w, scale_clamp, image_hw_dev,
                                                        score_thresh, cap, cand_boxes, cand_scores, cand_class, cand_roi, cand_count, cand_max);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// batched_nms helper: gather candidates into score order and add class * (max_coordinate + 1) to all four coordinates
__global__ void det_offset_gather_kernel(const float* __restrict__ cboxes, const int* __restrict__ cclass, const int* __restrict__ order,
                                         const int* __restrict__ ccount, const float* __restrict__ cmax, int cap, float* __restrict__ out) {
  int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  f32x4 v = {0, 0, 0, 0};
  if (i < ccount[b]) {
    int src = order[(size_t)b * cap + i];
    v = *reinterpret_cast<const f32x4*>(cboxes + ((size_t)b * cap + src) * 4);
    float off = (float)cclass[(size_t)b * cap + src] * (cmax[b] + 1.0f);
    v[0] += off; v[1] += off; v[2] += off; v[3] += off;
  }
  *reinterpret_cast<f32x4*>(out + ((size_t)b * cap + i) * 4) = v;
}
extern "C" int unit_detection_offset_gather(const float* cand_boxes, const int* cand_class, const int* order, const int* cand_count,
                                            const float* cand_max, int B, int cap, float* out, void* stream) {
  if (B == 0 || cap == 0) return UNIT_OK;
  det_offset_gather_kernel<<<dim3(cdiv(cap, 256), B), 256, 0, (hipStream_t)stream>>>(cand_boxes, cand_class, order, cand_count, cand_max, cap, out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// final gather: keep[j] indexes the score-sorted list; emit boxes (un-offset), scores, classes, roi indices
__global__ void det_final_kernel(const float* __restrict__ cboxes, const float* __restrict__ cscores, const int* __restrict__ cclass,
                                 const int* __restrict__ croi, const int* __restrict__ order, const int* __restrict__ keep,
                                 const int* __restrict__ keep_count, int cap, int topk, float* __restrict__ oboxes, float* __restrict__ oscores,
                                 int* __restrict__ oclass, int* __restrict__ oroi, int* __restrict__ ocount) {
  int b = blockIdx.x;
  int n = min(keep_count[b], topk);
  for (int j = threadIdx.x; j < topk; j += blockDim.x) {
    size_t o = (size_t)b * topk + j;
    if (j < n) {
      int src = order[(size_t)b * cap + keep[(size_t)b * topk + j]];
      size_t s = (size_t)b * cap + src;
      *reinterpret_cast<f32x4*>(oboxes + 4 * o) = *reinterpret_cast<const f32x4*>(cboxes + 4 * s);
      oscores[o] = cscores[s]; oclass[o] = cclass[s]; oroi[o] = croi[s];
    } else {
      *reinterpret_cast<f32x4*>(oboxes + 4 * o) = f32x4{0, 0, 0, 0};
      oscores[o] = 0.f; oclass[o] = -1; oroi[o] = -1;
    }
  }
  if (threadIdx.x == 0) ocount[b] = n;
}
extern "C" int unit_detection_finalize(const float* cand_boxes, const float* cand_scores, const int* cand_class, const int* cand_roi,
                                       const int* order, const int* keep, const int* keep_count, int B, int cap, int topk,
                                       float* out_boxes, float* out_scores, int* out_class, int* out_roi, int* out_count, void* stream) {
  if (B == 0) return UNIT_OK;
  det_final_kernel<<<B, 128, 0, (hipStream_t)stream>>>(cand_boxes, cand_scores, cand_class, cand_roi, order, keep, keep_count, cap, topk,
                                                      out_boxes, out_scores, out_class, out_roi, out_count);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// detector_postprocess (rcnn.py:411-429 -> detectron2): scale to the output resolution, clip, flag non-empty
__global__ void postprocess_kernel(float* __restrict__ boxes, const int* __restrict__ count, int topk, const float* __restrict__ scale_xy,
                                   const float* __restrict__ out_hw, unsigned char* __restrict__ nonempty) {
  int b = blockIdx.x;
  for (int j = threadIdx.x; j < topk; j += blockDim.x) {
    size_t o = (size_t)b * topk + j;
    f32x4 v = *reinterpret_cast<f32x4*>(boxes + 4 * o);
    float sx = scale_xy[2 * b], sy = scale_xy[2 * b + 1], oh = out_hw[2 * b], ow = out_hw[2 * b + 1];
    v[0] = fminf(fmaxf(v[0] * sx, 0.f), ow); v[2] = fminf(fmaxf(v[2] * sx, 0.f), ow);
    v[1] = fminf(fmaxf(v[1] * sy, 0.f), oh); v[3] = fminf(fmaxf(v[3] * sy, 0.f), oh);
    *reinterpret_cast<f32x4*>(boxes + 4 * o) = v;
    nonempty[o] = (j < count[b] && (v[2] - v[0]) > 0.f && (v[3] - v[1]) > 0.f) ? 1 : 0;
  }
}
extern "C" int unit_detector_postprocess(float* boxes, const int* count, int B, int topk, const float* scale_xy_dev, const float* out_hw_dev,
                                         unsigned char* nonempty, void* stream) {
  if (B == 0) return UNIT_OK;
  postprocess_kernel<<<B, 128, 0, (hipStream_t)stream>>>(boxes, count, topk, scale_xy_dev, out_hw_dev, nonempty);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// ---- output assembly without stock operators (modeling/inference.py) ---------------------------------------------------------------
// Stable compaction of the kept detections of every image (keep = j < count[b] [&& nonempty[b][j]]): rows of boxes / scores / classes
// (as int64, the reference's pred_classes dtype) / RoI indices / mask probabilities move to the front of their image's block, out_count[b]
// = rows kept. One workgroup per image; topk <= 1024. The reference does this with boolean-mask indexing per image and field
// (detectron2 detector_postprocess via meta_arch/rcnn.py:411-429): ~6 launches and a host sync per image.
__global__ void __launch_bounds__(256) compact_detections_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, const int* __restrict__ cls,
                                                                 const int* __restrict__ roi, const float* __restrict__ masks, int mask_elems,
                                                                 const int* __restrict__ count, const unsigned char* __restrict__ nonempty, int topk,
                                                                 float* __restrict__ oboxes, float* __restrict__ oscores, long* __restrict__ ocls,
                                                                 int* __restrict__ oroi, float* __restrict__ omasks, int* __restrict__ out_count) {
  __shared__ int pos[1024];
  __shared__ int total;
  const int b = blockIdx.x, c = min(count[b], topk);
  if (threadIdx.x == 0) {          // serial prefix over <= 1024 flags: a few hundred cycles, once per image
    int n = 0;
    for (int j = 0; j < topk; ++j) {
      bool keep = j < c && (nonempty == nullptr || nonempty[(size_t)b * topk + j]);
      pos[j] = keep ? n++ : -1;
    }
    total = n;
    out_count[b] = n;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < topk; j += blockDim.x) {
    int d = pos[j];
    if (d < 0) continue;
    size_t s = (size_t)b * topk + j, o = (size_t)b * topk + d;
    *reinterpret_cast<f32x4*>(oboxes + 4 * o) = *reinterpret_cast<const f32x4*>(boxes + 4 * s);
    oscores[o] = scores[s]; ocls[o] = (long)cls[s]; oroi[o] = roi[s];
  }
  if (masks != nullptr) {
    for (int j = 0; j < topk; ++j) {
      int d = pos[j];
      if (d < 0) continue;
      const float* src = masks + ((size_t)b * topk + j) * mask_elems;
      float* dst = omasks + ((size_t)b * topk + d) * mask_elems;
      for (int e = threadIdx.x; e < mask_elems; e += blockDim.x) dst[e] = src[e];
    }
  }
}
extern "C" int unit_compact_detections(const float* boxes, const float* scores, const int* cls, const int* roi, const float* masks, int mask_elems,
                                       const int* count, const unsigned char* nonempty, int B, int topk, float* oboxes, float* oscores,
                                       long* ocls, int* oroi, float* omasks, int* out_count, void* stream) {
  UNIT_CHECK_ARG(topk >= 0 && topk <= 1024, "compact_detections: topk <= 1024");
  UNIT_CHECK_ARG(masks == nullptr || (omasks != nullptr && omasks != masks), "compact_detections: masks need a separate output");
  if (B == 0) return UNIT_OK;
  compact_detections_kernel<<<B, 256, 0, (hipStream_t)stream>>>(boxes, scores, cls, roi, masks, mask_elems, count, nonempty, topk, oboxes, oscores, ocls,
                                                                oroi, omasks, out_count);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// boxes [B][T][4] -> RoIAlign rows [B*T][5] = (image index, x0, y0, x1, y1)
__global__ void boxes_to_rois5_kernel(const float* __restrict__ boxes, int T, long n, float* __restrict__ rois5) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(boxes + 4 * i);
  float* o = rois5 + 5 * i;
  o[0] = (float)(i / T); o[1] = v[0]; o[2] = v[1]; o[3] = v[2]; o[4] = v[3];
}
extern "C" int unit_boxes_to_rois5(const float* boxes, int B, int T, float* rois5, void* stream) {
  long n = (long)B * T;
  if (n == 0) return UNIT_OK;
  boxes_to_rois5_kernel<<<(unsigned)cdiv(n, 256L), 256, 0, (hipStream_t)stream>>>(boxes, T, n, rois5);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// out[b * T + j][:] = src[b * rcap + max(idx[b][j], 0)][:]   (rows of `row_bytes` bytes, a multiple of 4): similarity['seg'][filter_inds]
// of roi_heads.py:768-771, and any other per-image row gather
__global__ void gather_rows_kernel(const unsigned* __restrict__ src, const int* __restrict__ idx, int T, int rcap, int row_words, long n_rows,
                                   unsigned* __restrict__ out) {
  long r = blockIdx.x;
  if (r >= n_rows) return;
  long b = r / T;
  int k = idx[r];
  const unsigned* s = src + ((size_t)b * rcap + (k > 0 ? k : 0)) * row_words;
  unsigned* d = out + (size_t)r * row_words;
  for (int e = threadIdx.x; e < row_words; e += blockDim.x) d[e] = s[e];
}
extern "C" int unit_gather_rows(const void* src, const int* idx, int B, int T, int rcap, int row_bytes, void* out, void* stream) {
  UNIT_CHECK_ARG(row_bytes % 4 == 0, "gather_rows: rows of whole 32-bit words");
  long n = (long)B * T;
  if (n == 0 || row_bytes == 0) return UNIT_OK;
  UNIT_CHECK_ARG(n < 0x7FFFFFFFl, "gather_rows: too many rows");
  gather_rows_kernel<<<(unsigned)n, 128, 0, (hipStream_t)stream>>>((const unsigned*)src, idx, T, rcap, row_bytes / 4, n, (unsigned*)out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

// the first `take` rows of each of `nb` blocks of `block_rows` rows -> one dense [nb * take][row] tensor (rows of `row_bytes` bytes, a multiple
// of 4): the foreground RoI slots of every image for the mask head (roi_heads.py:691-710 select_foreground_proposals; the sampler emits
// [fg..., bg...] per image). Replaces one torch.cat of per-image slices per field.
__global__ void gather_blocks_kernel(const unsigned* __restrict__ src, int block_rows, int take, int row_words, unsigned* __restrict__ out) {
  long r = blockIdx.x;                      // output row
  long b = r / take, j = r - b * take;
  const unsigned* s = src + ((size_t)b * block_rows + j) * row_words;
  unsigned* d = out + (size_t)r * row_words;
  if ((row_words & 3) == 0) {
    for (int e = threadIdx.x; e < row_words / 4; e += blockDim.x) reinterpret_cast<u32x4*>(d)[e] = reinterpret_cast<const u32x4*>(s)[e];
  } else {
    for (int e = threadIdx.x; e < row_words; e += blockDim.x) d[e] = s[e];
  }
}
extern "C" int unit_gather_blocks(const void* src, int nb, int block_rows, int take, int row_bytes, void* out, void* stream) {
  UNIT_CHECK_ARG(row_bytes % 4 == 0 && take <= block_rows, "gather_blocks: rows of whole 32-bit words, take <= block_rows");
  UNIT_CHECK_ARG((row_bytes % 16 != 0) || (((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0)), "gather_blocks: 16B alignment");
  long n = (long)nb * take;
  if (n == 0 || row_bytes == 0) return UNIT_OK;
  gather_blocks_kernel<<<(unsigned)n, row_bytes >= 4096 ? 256 : 64, 0, (hipStream_t)stream>>>((const unsigned*)src, block_rows, take, row_bytes / 4, (unsigned*)out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
