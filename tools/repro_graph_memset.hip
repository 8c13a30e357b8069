// Standalone check (no torch, no libmfhip): is a hipMemsetAsync captured into a hipGraph ordered against the kernels
// around it when the graph is replayed?   hipcc --offload-arch=gfx950 tools/repro_graph_memset.hip -o /tmp/repro && /tmp/repro
//
// Graph (stream capture, one stream):  memset(buf, VALUE) -> k_bump(buf) -> k_check(buf, expect) [-> k_scribble(buf)]
// k_bump adds 1 to every word, k_check counts the words that are not VALUE_WORD + 1, k_scribble leaves garbage behind
// for the NEXT replay's memset to clear.  Any non-zero count = the memset node did not run (completely) between the
// previous replay's scribble and this replay's bump.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_bump(uint32_t *p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1u;
}
__global__ void k_check(const uint32_t *p, int64_t n, uint32_t expect, unsigned long long *bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && p[i] != expect) atomicAdd(bad, 1ull);
}
__global__ void k_scribble(uint32_t *p, int64_t n, uint32_t salt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0x5a5a0000u + salt + (uint32_t)i;
}

static int run(int64_t bytes, int value, int replays, bool as_graph) {
  const int64_t n = bytes / 4;
  uint32_t *buf;
  unsigned long long *bad, h_bad = 0;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&bad, 8));
  CK(hipMemset(bad, 0, 8));
  CK(hipMemset(buf, 0x77, bytes));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const uint32_t word = (uint32_t)(value & 0xff) * 0x01010101u;
  const unsigned nb = (unsigned)((n + 255) / 256);
  auto enqueue = [&]() -> int {
    CK(hipMemsetAsync(buf, value, bytes, s));
    hipLaunchKernelGGL(k_bump, dim3(nb), dim3(256), 0, s, buf, n);
    hipLaunchKernelGGL(k_check, dim3(nb), dim3(256), 0, s, buf, n, word + 1u, bad);
    hipLaunchKernelGGL(k_scribble, dim3(nb), dim3(256), 0, s, buf, n, 7u);
    return 0;
  };
  if (as_graph) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    if (enqueue()) return 2;
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < replays; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  } else {
    for (int r = 0; r < replays; ++r)
      if (enqueue()) return 2;
    CK(hipStreamSynchronize(s));
  }
  CK(hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost));
  printf("{\"mode\": \"%s\", \"bytes\": %lld, \"value\": %d, \"replays\": %d, \"bad_words\": %llu}\n",
         as_graph ? "graph" : "stream", (long long)bytes, value, replays, h_bad);
  CK(hipFree(buf));
  CK(hipFree(bad));
  CK(hipStreamDestroy(s));
  return h_bad ? 1 : 0;
}

// The pattern of csrc/sparseconv.hip before round 4: THREE consecutive memsets into one carved-up workspace
// (8 counters, per-voxel counts = 0, per-voxel chain heads = -1), then a kernel that links points into the chains
// with atomics, a kernel that walks every chain and counts out-of-range steps, and ~40 filler kernel nodes in front
// (the real graph has ~60 kernel nodes before the sparse convolution).
__global__ void k_link(const int *vox, int n, int *counts, int *head, int *link, int *cls) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  atomicAdd(&counts[vox[i]], 1);
  link[i] = atomicExch(&head[vox[i]], i);
  atomicAdd(&cls[vox[i] & 7], 1);
}
__global__ void k_walk(const int *vox, int n, const int *counts, const int *head, const int *link, const int *cls,
                       unsigned long long *bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    int s = 0;
    for (int c = 0; c < 8; ++c) s += cls[c];
    if (s != n) atomicAdd(bad, 1ull);
  }
  if (i >= n || head[vox[i]] != i) return;
  int m = i;
  for (int k = 0; k < counts[vox[i]]; ++k) {
    if (m < 0 || m >= n) { atomicAdd(bad, 1ull); return; }   // (the real kernel dereferences values[m * ld])
    m = link[m];
  }
  if (m != -1) atomicAdd(bad, 1ull);
}
__global__ void k_filler(float *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
__global__ void k_new_frame(int *vox, int n, int V, unsigned seed) {  // a new point set per replay
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) vox[i] = (int)(((unsigned)i * 2654435761u + seed * 40503u) % (unsigned)V);
}

static int run_chain(int replays, bool as_graph, int fillers) {
  const int V = 32768, n = 1000;
  char *ws;
  int *vox, *link;
  float *fill;
  unsigned long long *bad, h_bad = 0;
  CK(hipMalloc(&ws, 512 + 2 * V * 4));
  CK(hipMalloc(&vox, n * 4));
  CK(hipMalloc(&link, n * 4));
  CK(hipMalloc(&fill, 1 << 22));
  CK(hipMalloc(&bad, 8));
  CK(hipMemset(bad, 0, 8));
  CK(hipMemset(fill, 0, 1 << 22));
  CK(hipMemset(ws, 0x33, 512 + 2 * V * 4));
  int *cls = (int *)ws, *counts = (int *)(ws + 512), *head = (int *)(ws + 512 + V * 4);
  hipStream_t s;
  CK(hipStreamCreate(&s));
  auto enqueue = [&]() -> int {
    for (int f = 0; f < fillers; ++f) hipLaunchKernelGGL(k_filler, dim3(4096), dim3(256), 0, s, fill, 1 << 20);
    CK(hipMemsetAsync(cls, 0, 512, s));
    CK(hipMemsetAsync(counts, 0, V * 4, s));
    CK(hipMemsetAsync(head, 0xff, V * 4, s));
    hipLaunchKernelGGL(k_link, dim3((n + 255) / 256), dim3(256), 0, s, vox, n, counts, head, link, cls);
    hipLaunchKernelGGL(k_walk, dim3((n + 255) / 256), dim3(256), 0, s, vox, n, counts, head, link, cls, bad);
    return 0;
  };
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  if (as_graph) {
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    if (enqueue()) return 2;
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  }
  for (int r = 0; r < replays; ++r) {
    hipLaunchKernelGGL(k_new_frame, dim3((n + 255) / 256), dim3(256), 0, s, vox, n, V, (unsigned)r);
    if (as_graph) { CK(hipGraphLaunch(ge, s)); } else if (enqueue()) return 2;
  }
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost));
  printf("{\"mode\": \"%s\", \"pattern\": \"3 memsets -> link -> walk\", \"fillers\": %d, \"replays\": %d, \"bad\": %llu}\n",
         as_graph ? "graph" : "stream", fillers, replays, h_bad);
  return h_bad ? 1 : 0;
}

int main() {
  int rc = 0;
  const int64_t sizes[] = {512, 4096, 131072, 1 << 20, 8 << 20};
  for (int as_graph = 0; as_graph < 2; ++as_graph)
    for (int64_t b : sizes)
      for (int value : {0, 0xff}) rc |= run(b, value, 300, as_graph != 0);
  for (int as_graph = 0; as_graph < 2; ++as_graph)
    for (int fillers : {0, 40}) rc |= run_chain(400, as_graph != 0, fillers);
  printf("{\"any_violation\": %s}\n", rc ? "true" : "false");
  return 0;
}
