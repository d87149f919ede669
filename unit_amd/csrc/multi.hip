// multi.hip -- multi-tensor end-of-step kernels: ONE launch over all trainable conv weights instead of one per layer.
//   unit_multi_wgrad_reduce : grads[off+i] = scale[k] * sum_s partial_s[i]      (ordered split-M slab reduction + FrozenBN fold)
//   unit_multi_weight_prep  : bf16/fp32 forward copy [K][R][S][C] (scale folded) + flipped/transposed dgrad copy [C][R][S][K]
// The per-layer launches they replace (~110 wgrad_reduce + ~110 weight_prep of 5-10 us each) were launch-bound.
#include "common.h"

struct UnitTensorDesc {
  const float* partial;
  const float* scale;
  void* wf;
  void* wd;
  long offset;
  int splits, K, R, S, C, block0;
  // weight prep only: row pitches (elements) of the two copies, 0 = contiguous. A pitched copy is one operand of a concatenated GEMM weight:
  // wf rows [k] of pitch ldwf (1x1 convs: [K][C_a + C_b], the dual-input conv3 + shortcut GEMM of a Res5 head's first block), wd rows
  // [(c, r', s')] of pitch ldwd ([C][K_a + K_b], its dgrad counterpart). wf may be NULL (only the dgrad copy is wanted).
  int ldwf, ldwd;
};

#define MT_ELEMS_PER_BLOCK 1024

__device__ __forceinline__ int find_tensor(const UnitTensorDesc* __restrict__ d, int n, int b) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (d[mid].block0 <= b) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ void multi_reduce_kernel(const UnitTensorDesc* __restrict__ descs, int n, float* __restrict__ grads) {
  int t = find_tensor(descs, n, blockIdx.x);
  UnitTensorDesc d = descs[t];
  long KK = (long)d.K * d.R * d.S * d.C;
  long i = (long)(blockIdx.x - d.block0) * MT_ELEMS_PER_BLOCK + threadIdx.x * 4;
  if (i >= KK || d.partial == nullptr) return;
  int nsp = d.splits < 0 ? -d.splits : d.splits;      // splits < 0: second contribution to the same gradient -> accumulate
  f32x4 s = *reinterpret_cast<const f32x4*>(d.partial + i);
  for (int sp = 1; sp < nsp; ++sp) s += *reinterpret_cast<const f32x4*>(d.partial + (size_t)sp * KK + i);
  if (d.scale) s *= d.scale[i / ((long)d.R * d.S * d.C)];
  if (d.splits < 0) s += *reinterpret_cast<const f32x4*>(grads + d.offset + i);
  *reinterpret_cast<f32x4*>(grads + d.offset + i) = s;
}

__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
  bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  *reinterpret_cast<bf16x4*>(p) = o;
}

// One workgroup = one 32 (k) x 32 (c) tile of one (r, s) tap: the forward copy keeps the [K][R][S][C] order (coalesced
// along c); the dgrad copy is [C][R'][S'][K], i.e. a transpose of the tile, staged through LDS so that its stores are
// coalesced along k as well (the direct form scattered 2-byte stores with a stride of K elements).
// block count per tensor: cdiv(K,32) * cdiv(C,32) * R * S  (unit_amd/multi.py builds block0 with the same formula).
template <typename T>
__global__ void __launch_bounds__(256) multi_prep_kernel(const UnitTensorDesc* __restrict__ descs, int n, const float* __restrict__ params) {
  __shared__ float tile[32][33];
  int t = find_tensor(descs, n, blockIdx.x);
  UnitTensorDesc d = descs[t];
  int lb = blockIdx.x - d.block0;
  int tiles_c = (d.C + 31) >> 5, tiles_k = (d.K + 31) >> 5;
  int rs = lb / (tiles_k * tiles_c); int rem = lb - rs * (tiles_k * tiles_c);
  int k0 = (rem / tiles_c) * 32, c0 = (rem % tiles_c) * 32;
  int r = rs / d.S, sx = rs - r * d.S;
  T* wf = (T*)d.wf; T* wd = (T*)d.wd;
  {
    int ky = threadIdx.x >> 3, cx = (threadIdx.x & 7) * 4;
    int k = k0 + ky, c = c0 + cx;
    f32x4 p = {0.f, 0.f, 0.f, 0.f};
    if (k < d.K && c < d.C) {                                  // C % 4 == 0: the 4 channels are in range together
      size_t i = ((size_t)k * d.R * d.S + rs) * d.C + c;
      p = *reinterpret_cast<const f32x4*>(params + d.offset + i);
      float sc = d.scale ? d.scale[k] : 1.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] *= sc;
      if (wf) store4(wf + (d.ldwf ? (size_t)k * d.ldwf + (size_t)rs * d.C + c : i), p);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[cx + j][ky] = p[j];
  }
  if (!wd) return;
  __syncthreads();
  {
    int cy = threadIdx.x >> 3, kx = (threadIdx.x & 7) * 4;
    int c = c0 + cy;
    if (c < d.C) {
      size_t o = (((size_t)c * d.R + (d.R - 1 - r)) * d.S + (d.S - 1 - sx)) * (d.ldwd ? d.ldwd : d.K) + k0 + kx;
      if ((d.K & 3) == 0 && (d.ldwd & 3) == 0) {                                      // rows of the dgrad copy stay 4-element aligned
        if (k0 + kx < d.K) store4(wd + o, f32x4{tile[cy][kx], tile[cy][kx + 1], tile[cy][kx + 2], tile[cy][kx + 3]});
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k0 + kx + j < d.K) wd[o + j] = (T)tile[cy][kx + j];
      }
    }
  }
}

extern "C" size_t unit_tensor_desc_bytes(void) { return sizeof(UnitTensorDesc); }

extern "C" int unit_multi_wgrad_reduce(const void* descs_dev, int n, int total_blocks, float* grads_flat, void* stream) {
  if (n == 0 || total_blocks == 0) return UNIT_OK;
  multi_reduce_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>((const UnitTensorDesc*)descs_dev, n, grads_flat);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}

extern "C" int unit_multi_weight_prep(const void* descs_dev, int n, int total_blocks, const float* params_flat, int dtype, void* stream) {
  if (n == 0 || total_blocks == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UNIT_BF16) multi_prep_kernel<bf16_t><<<total_blocks, 256, 0, st>>>((const UnitTensorDesc*)descs_dev, n, params_flat);
  else multi_prep_kernel<float><<<total_blocks, 256, 0, st>>>((const UnitTensorDesc*)descs_dev, n, params_flat);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// unit_stream_wait_stream: make everything enqueued on `waiter` from now on wait for everything enqueued on `signaller` so far
// (hipEventRecord + hipStreamWaitEvent on an event of a small ring). The step forks ~100 weight-gradient launches per step onto a
// side stream; through torch.cuda.Event / Stream objects each fork cost the host ~25 us of Python (a quarter of the step's enqueue
// time), this is one C call. Works inside a stream capture (the record / wait pair becomes a graph edge). One process per GPU,
// called from one thread; an event is reused only 1024 forks later.
static hipEvent_t g_fork_events[1024];
static bool g_fork_events_ready = false;
static unsigned g_fork_next = 0;

extern "C" int unit_stream_wait_stream(void* waiter, void* signaller) {
  if (waiter == signaller) return UNIT_OK;
  if (!g_fork_events_ready) {
    for (int i = 0; i < 1024; ++i) {
      hipError_t e = hipEventCreateWithFlags(&g_fork_events[i], hipEventDisableTiming);
      if (e != hipSuccess) { unit_set_error(hipGetErrorString(e)); return UNIT_ERR_LAUNCH; }
    }
    g_fork_events_ready = true;
  }
  hipEvent_t ev = g_fork_events[g_fork_next++ & 1023];
  hipError_t e = hipEventRecord(ev, (hipStream_t)signaller);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
  if (e != hipSuccess) { unit_set_error(hipGetErrorString(e)); return UNIT_ERR_LAUNCH; }
  return UNIT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// unit_shard_sum: out[i] = parts[0][i] + parts[1][i] + ... + parts[nparts-1][i], added in that (rank) order in fp32.
// The local half of the "direct" gradient exchange of unit_amd/parallel.py (all-to-all of the ranks' shard contributions over
// the seven xGMI links at once -> this ordered sum on the shard's owner -> all-gather): every element is summed by ONE rank in a
// fixed order, so all ranks end with bit-identical gradients and a re-run reproduces them. parts = [nparts][n] contiguous,
// fp32 or bf16 (widened before the add); out fp32. HBM-bound: (nparts + 1) * n * 4 bytes, 16 bytes per lane.
template <typename T>
__global__ void __launch_bounds__(256) shard_sum_kernel(const T* __restrict__ parts, int nparts, long n, float* __restrict__ out) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    f32x4 s;
    {
      const T* p = parts + i;
      s = f32x4{(float)p[0], (float)p[1], (float)p[2], (float)p[3]};
    }
    for (int r = 1; r < nparts; ++r) {
      const T* p = parts + (size_t)r * n + i;
      s += f32x4{(float)p[0], (float)p[1], (float)p[2], (float)p[3]};
    }
    out[i] = s[0]; out[i + 1] = s[1]; out[i + 2] = s[2]; out[i + 3] = s[3];
  } else {
    for (long j = i; j < n; ++j) {
      float s = (float)parts[j];
      for (int r = 1; r < nparts; ++r) s += (float)parts[(size_t)r * n + j];
      out[j] = s;
    }
  }
}

extern "C" int unit_shard_sum(const void* parts, int dtype, int nparts, long n, float* out, void* stream) {
  UNIT_CHECK_ARG(nparts >= 1 && n >= 0, "shard_sum: nparts >= 1");
  if (n == 0) return UNIT_OK;
  hipStream_t st = (hipStream_t)stream;
  int g = cdiv(cdiv(n, 4), 256);
  if (dtype == UNIT_BF16) shard_sum_kernel<bf16_t><<<g, 256, 0, st>>>((const bf16_t*)parts, nparts, n, out);
  else shard_sum_kernel<float><<<g, 256, 0, st>>>((const float*)parts, nparts, n, out);
  UNIT_LAUNCH_CHECK();
  return UNIT_OK;
}
