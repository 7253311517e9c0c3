// Word-embedding gradient: dword[ids[t]] += dx[t] for T tokens — the transpose of the gather in BertEmbeddings
// (HF modeling_bert; reached from `CXRBertModel.forward`, health_multimodal/text/model/modelling_cxrbert.py:70-99).
//
// Deterministic, no floating-point atomics.  Tokens are sorted by (id, position) with a radix sort on the unique keys
// (id << 32 | t); the sorted list is cut into chunks of CH entries, one workgroup each:
//   * a run of equal ids that lies strictly inside a chunk belongs to that chunk alone: it is summed in position order and added
//     to its embedding row directly;
//   * the first run of a chunk (it may continue the previous chunk's last run) and the last run (it may continue into the next
//     chunk) are written as partial rows; a second pass, one workgroup per chunk, finishes every run that STARTS in its chunk by
//     adding the partials of the chunks it spans in order ([PAD] and [CLS] rows span hundreds of chunks).
// Every row is therefore a sum in ascending token position, whatever the scheduling: two runs give identical bits.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "cxrk.h"
#include "cxrk_common.h"

using namespace cxrk;

namespace {
constexpr int CH = 32;   // sorted entries per chunk

__global__ void make_keys_kernel(const long* __restrict__ ids, long T, unsigned long long* __restrict__ keys) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < T) keys[t] = ((unsigned long long)ids[t] << 32) | (unsigned long long)t;
}

// meta[c] = {number of runs (1: the whole chunk is one run; else >= 2), id of the first run, id of the last run}
__global__ __launch_bounds__(256) void chunk_sum_kernel(const unsigned long long* __restrict__ skeys, const float* __restrict__ dx, long T,
                                                        int H, float* __restrict__ dword, int* __restrict__ meta,
                                                        float* __restrict__ part_head, float* __restrict__ part_tail) {
  const long c = blockIdx.x;
  const long p0 = c * CH, p1 = min(T, p0 + (long)CH);
  __shared__ unsigned long long sk[CH];
  for (int i = threadIdx.x; i < CH; i += blockDim.x) sk[i] = p0 + i < p1 ? skeys[p0 + i] : 0ull;
  __syncthreads();
  const int n = (int)(p1 - p0);
  int nruns = 1;
  for (int i = 1; i < n; ++i) nruns += (sk[i] >> 32) != (sk[i - 1] >> 32);
  for (int col = threadIdx.x; col < H; col += blockDim.x) {
    float acc = 0.f;
    int run = 0;
    for (int i = 0; i < n; ++i) {
      const long id = (long)(sk[i] >> 32), t = (long)(sk[i] & 0xffffffffull);
      acc += dx[t * H + col];
      const bool last_of_run = i + 1 == n || (long)(sk[i + 1] >> 32) != id;
      if (last_of_run) {
        if (run == 0) part_head[c * H + col] = acc;                       // first run: may continue the previous chunk
        else if (i + 1 == n) part_tail[c * H + col] = acc;                // last run (of >= 2): may continue into the next chunk
        else dword[id * H + col] += acc;                                   // interior run: this chunk owns the row
        acc = 0.f; ++run;
      }
    }
  }
  if (threadIdx.x == 0) { meta[3 * c] = nruns; meta[3 * c + 1] = (int)(sk[0] >> 32); meta[3 * c + 2] = (int)(sk[n - 1] >> 32); }
}

// finishes every boundary run that starts in chunk c: its pieces are part_tail / part_head of c and part_head of the following
// chunks while they continue the same id
__global__ __launch_bounds__(256) void boundary_kernel(const int* __restrict__ meta, long nchunks, int H, const float* __restrict__ part_head,
                                                       const float* __restrict__ part_tail, float* __restrict__ dword) {
  const long c = blockIdx.x;
  const int nruns = meta[3 * c], head_id = meta[3 * c + 1], tail_id = meta[3 * c + 2];
  for (int which = 0; which < 2; ++which) {
    int id; const float* first;
    bool reaches_end;
    if (which == 0) {   // the first run, if it does not continue the previous chunk
      if (c > 0 && meta[3 * (c - 1) + 2] == head_id) continue;
      id = head_id; first = part_head + c * H; reaches_end = nruns == 1;
    } else {            // the last run of a chunk with >= 2 runs starts inside it
      if (nruns < 2) continue;
      id = tail_id; first = part_tail + c * H; reaches_end = true;
    }
    long last = c;     // last chunk the run covers
    if (reaches_end)
      while (last + 1 < nchunks && meta[3 * (last + 1) + 1] == id) { ++last; if (meta[3 * last] > 1) break; }
    for (int col = threadIdx.x; col < H; col += blockDim.x) {
      float acc = first[col];
      for (long cc = c + 1; cc <= last; ++cc) acc += part_head[cc * H + col];
      dword[(long)id * H + col] += acc;
    }
  }
}

size_t sort_temp_bytes(long T) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)T);
  return (bytes + 255) & ~(size_t)255;
}
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace

extern "C" size_t cxrk_embed_bwd_ws_bytes(long T, int H) {
  const long nchunks = (T + CH - 1) / CH;
  return 2 * align256((size_t)T * 8) + sort_temp_bytes(T) + align256((size_t)nchunks * 3 * sizeof(int)) +
         2 * align256((size_t)nchunks * H * sizeof(float));
}

extern "C" int cxrk_embed_bwd(const long* ids, const float* dx, long T, int H, float* dword, float* ws, size_t ws_bytes,
                              hipStream_t stream) {
  CXRK_CHECK_ARG(ids && dx && dword && T > 0 && H > 0 && T < (1L << 31));
  if (ws == nullptr || ws_bytes < cxrk_embed_bwd_ws_bytes(T, H)) return CXRK_ERR_WS;
  const long nchunks = (T + CH - 1) / CH;
  unsigned char* p = reinterpret_cast<unsigned char*>(ws);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(p); p += align256((size_t)T * 8);
  unsigned long long* skeys = reinterpret_cast<unsigned long long*>(p); p += align256((size_t)T * 8);
  void* temp = p; size_t temp_bytes = sort_temp_bytes(T); p += temp_bytes;
  int* meta = reinterpret_cast<int*>(p); p += align256((size_t)nchunks * 3 * sizeof(int));
  float* part_head = reinterpret_cast<float*>(p); p += align256((size_t)nchunks * H * sizeof(float));
  float* part_tail = reinterpret_cast<float*>(p);
  hipLaunchKernelGGL(make_keys_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, stream, ids, T, keys);
  CXRK_LAUNCH_CHECK();
  if (hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, keys, skeys, (int)T, 0, 64, stream) != hipSuccess) return CXRK_ERR_LAUNCH;
  hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)nchunks), dim3(256), 0, stream, skeys, dx, T, H, dword, meta, part_head, part_tail);
  CXRK_LAUNCH_CHECK();
  hipLaunchKernelGGL(boundary_kernel, dim3((unsigned)nchunks), dim3(256), 0, stream, meta, nchunks, H, part_head, part_tail, dword);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
