// Tensor-parallel collectives over xGMI for decode-sized messages, hand-written for gfx950.
// Replaces dist.all_reduce after the row-parallel GEMMs (nano-vllm layers/linear.py:153-156), after the
// vocab-parallel embedding (layers/embed_head.py:41) and dist.gather of the logits (embed_head.py:62-65; here
// only 8 bytes per row travel: the sampler's per-shard winner), where the reference goes through NCCL.
//
// Why not RCCL for these: a decode step of Qwen3-32B makes 129 all-reduces of <= 2.6 MB; a ring over the
// point-to-point xGMI links pays 2*(W-1) hops of latency each. MI355X GPUs are FULLY connected (7 links per
// GPU), so every rank can read every peer directly: one kernel per collective, every rank's buffer mapped into
// every process with hipIpc, flags instead of a communicator. The launch is enqueue-only on the caller's stream
// (hipGraph-capturable: epochs live in device memory and advance by themselves on every replay).
//
// Protocol (per call; workgroup b of every rank runs the same protocol instance b, so there is no grid barrier):
//   phase 0  workgroup b copies ITS rows of the local partial-sum tensor into this rank's shared `data` region,
//            then publishes flag0[b][rank] = epoch in every peer's flag region.
//   phase 1  (two-shot) rank r owns the column slice [r*H/W, (r+1)*H/W) of every row: workgroup b waits for
//            flag0[b][*], sums that slice of its rows over all W ranks' `data` regions in rank order (fp32, one
//            rounding to bf16 — every rank later reads the SAME bf16 values: results are identical on all ranks
//            and run-to-run), writes it to this rank's `reduced` region and publishes flag1[b][rank].
//   phase 2  workgroup b waits for flag1[b][*] and assembles its full rows from the W owners' `reduced` regions;
//            epilogue = plain store, or the reference's add_rms_forward (layers/layernorm.py:28-40: residual add
//            + RMSNorm) so the all-reduce and the norm that always follows it are ONE launch.
//   One-shot (tiny messages): phase 1 is skipped and phase 2 sums the peers' `data` rows directly; a closing flag1
//            exchange then tells every rank that all peers are done reading its rows.
// Re-use safety: a rank overwrites `data` in call e+1 only after its kernel of call e has completed, i.e. after
// every one of its workgroups has seen flag1 = e from every peer, which peers publish after their phase-1 reads;
// it overwrites `reduced` rows in phase 1 of call e+1 only after flag0 = e+1 from every peer, which a peer
// publishes from its kernel e+1, i.e. after its kernel e (all phase-2 reads) has completed.
// Visibility: the shared buffer is allocated UNCACHED (hipDeviceMallocUncached: no L2/L1 residency on either
// side of a link) and every hand-off is {per-wave vmcnt(0) drain, barrier, system-scope release, flag store} ->
// {relaxed system-scope poll, system-scope acquire, barrier}, the cross-device form of cdna_hip_programming.md
// Guideline 16. Every spin is bounded; a timeout is latched in the flag region and reported by
// nvl_allreduce_status() (the result of that call is then garbage, but nothing hangs).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int kMaxWorld = 8;
constexpr int kMaxBlocks = 256;
constexpr size_t kFlagBytes = 64 * 1024;

struct Flags {                                  // lives at offset 0 of every rank's shared buffer
  uint32_t flag0[kMaxBlocks][kMaxWorld];        // written by peers (and by the owner itself for its own column)
  uint32_t flag1[kMaxBlocks][kMaxWorld];
  uint32_t epoch[kMaxBlocks];                   // local: last completed call of protocol instance b
  uint32_t error;                               // local: latched spin timeout
};
static_assert(sizeof(Flags) <= kFlagBytes, "flag region too small");

struct CommDev {                                // passed by value to the kernels
  unsigned char* base[kMaxWorld];
  int rank, world;
  uint64_t data_off, red_off;
  int fences;                                   // 1: system-scope release / acquire around every hand-off (default)
};

struct Comm {                                   // host-side object behind the opaque handle
  CommDev dev;
  size_t data_bytes, total_bytes;
  bool connected;
  bool opened[kMaxWorld];
};

__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Tens of seconds with the s_sleep below: generous, because several ranks may be time-sliced on ONE GPU in the
// single-GPU functional tests; on a healthy 8-GPU node a wait is microseconds.
constexpr uint32_t kSpinLimit = 40u * 1000u * 1000u;

// Every thread of the workgroup calls both. `which` = 0 / 1 selects flag0 / flag1.
// c.fences == 0 ("lean": selected by tp.init_p2p only after a randomised stress self-check of BOTH flavours has passed
// on the topology at hand; NVL_TP_P2P_HANDOFF=fenced / lean forces one): everything a peer reads
// lives in UNCACHED memory, so a store that has been acknowledged (vmcnt) is in memory and a load cannot hit a stale
// line: the per-wave drain + barrier orders payload before flag — the "write-through payload -> vmcnt(0) -> flag" form
// of Guideline 16 (R1) — without the L2 write-back / invalidate of a system-scope fence, four of which per call were
// most of the protocol cost (131 x 5120, 2 ranks on one GPU: 14.7 -> 10.4 us; with the producer GEMM writing into the
// shared region, 8.7 us; profiles/r02_p2p_bench_w2.json). Exactness tests run both flavours (tests/test_tp_gpu.py).
__device__ __forceinline__ void publish(const CommDev& c, int b, int which, uint32_t epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its own stores
  __syncthreads();
  if (threadIdx.x < 64) {
    if (c.fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the compiler may drop the fence's own wait)
    if ((int)threadIdx.x < c.world) {
      Flags* f = reinterpret_cast<Flags*>(c.base[threadIdx.x]);
      st_sys(which == 0 ? &f->flag0[b][c.rank] : &f->flag1[b][c.rank], epoch);
    }
  }
}
__device__ __forceinline__ void await(const CommDev& c, int b, int which, uint32_t epoch) {
  if (threadIdx.x < 64) {
    Flags* f = reinterpret_cast<Flags*>(c.base[c.rank]);
    if ((int)threadIdx.x < c.world) {
      const uint32_t* w = which == 0 ? &f->flag0[b][threadIdx.x] : &f->flag1[b][threadIdx.x];
      uint32_t spins = 0;
      // epochs only grow; (int) difference tolerates wrap-around
      while ((int)(ld_sys(w) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > kSpinLimit) {
          st_sys(&f->error, 1u);
          break;
        }
      }
    }
    if (c.fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the poll's loads have returned before anyone reads on
  }
  __syncthreads();
}

enum { EPI_STORE = 0, EPI_ADD_RMSNORM = 1 };

// rows x hidden bf16, hidden % (8 * world) == 0, hidden <= 256 * 8 * kMaxChunks
template <int EPI, bool ONESHOT>
__global__ __launch_bounds__(256) void allreduce_rows_kernel(CommDev c, const bf16_t* __restrict__ in,
                                                              bf16_t* __restrict__ out, bf16_t* __restrict__ residual,
                                                              const bf16_t* __restrict__ weight, int rows, int hidden,
                                                              float eps) {
  constexpr int kMaxChunks = 4;
  __shared__ float red[4];
  const int b = blockIdx.x, nb = gridDim.x, tid = threadIdx.x;
  Flags* mine = reinterpret_cast<Flags*>(c.base[c.rank]);
  const uint32_t epoch = ld_sys(&mine->epoch[b]) + 1u;
  const int nchunks = hidden >> 3;                         // 16-byte chunks per row
  const int slice = nchunks / c.world;                     // chunks per owner column slice

  // ---- phase 0: my rows of the local partial sums -> my shared data region (skipped when the producer — the
  //      row-parallel GEMM — already wrote them there: nvl_allreduce_buffer) ----------------------------------------
  bf16_t* data = reinterpret_cast<bf16_t*>(c.base[c.rank] + c.data_off);
  if (in != data) {
    for (int row = b; row < rows; row += nb)
      for (int ch = tid; ch < nchunks; ch += 256) {
        const int64_t at = (int64_t)row * hidden + ch * 8;
        *reinterpret_cast<u32x4_t*>(data + at) = *reinterpret_cast<const u32x4_t*>(in + at);
      }
  }
  publish(c, b, 0, epoch);
  await(c, b, 0, epoch);

  if constexpr (!ONESHOT) {
    // ---- phase 1: reduce my column slice of my rows over all ranks (fixed rank order) -----------------------
    bf16_t* reduced = reinterpret_cast<bf16_t*>(c.base[c.rank] + c.red_off);
    for (int row = b; row < rows; row += nb)
      for (int ch = tid; ch < slice; ch += 256) {
        const int64_t at = (int64_t)row * hidden + (c.rank * slice + ch) * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < c.world; ++p) {
          float v[8];
          unpack8(*reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(c.base[p] + c.data_off) + at), v);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
        *reinterpret_cast<u32x4_t*>(reduced + at) = pack8(acc);
      }
    publish(c, b, 1, epoch);
    await(c, b, 1, epoch);
  }

  // ---- phase 2: assemble full rows (+ epilogue) --------------------------------------------------------------
  for (int row = b; row < rows; row += nb) {
    float v[kMaxChunks][8];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k) {
      const int ch = tid + k * 256;
      if (ch < nchunks) {
        const int64_t at = (int64_t)row * hidden + ch * 8;
        if constexpr (ONESHOT) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[k][i] = 0.f;
          for (int p = 0; p < c.world; ++p) {
            float t[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(c.base[p] + c.data_off) + at), t);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[k][i] += t[i];
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) v[k][i] = round_bf16(v[k][i]);      // same rounding point as two-shot
        } else {
          const int owner = ch / slice;
          unpack8(*reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(c.base[owner] + c.red_off) + at),
                  v[k]);
        }
        if constexpr (EPI == EPI_STORE) {
          *reinterpret_cast<u32x4_t*>(out + at) = pack8(v[k]);
        } else {
          float r[8];
          unpack8(*reinterpret_cast<const u32x4_t*>(residual + at), r);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            v[k][i] += r[i];                                   // un-rounded fp32 sum feeds the norm (layernorm.py:35-38)
            ss += v[k][i] * v[k][i];
          }
          *reinterpret_cast<u32x4_t*>(residual + at) = pack8(v[k]);
        }
      }
    }
    if constexpr (EPI == EPI_ADD_RMSNORM) {
      ss = wave_allreduce_sum(ss);
      __syncthreads();                                         // `red` may still be read by the previous row
      if ((tid & 63) == 0) red[tid >> 6] = ss;
      __syncthreads();
      const float tot = red[0] + red[1] + red[2] + red[3];
      const float rstd = rsqrtf(tot / (float)hidden + eps);
#pragma unroll
      for (int k = 0; k < kMaxChunks; ++k) {
        const int ch = tid + k * 256;
        if (ch < nchunks) {
          float wf[8], o[8];
          unpack8(*reinterpret_cast<const u32x4_t*>(weight + ch * 8), wf);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = v[k][i] * rstd * wf[i];
          *reinterpret_cast<u32x4_t*>(out + (int64_t)row * hidden + ch * 8) = pack8(o);
        }
      }
    }
  }
  if constexpr (ONESHOT) {
    // One-shot has no second exchange of its own, so nothing yet says "every peer has finished reading my rows":
    // without this hand-shake a fast rank's NEXT producer (the GEMM writing into its shared region, or the next
    // call's copy-in) could overwrite rows a slower peer is still summing.
    publish(c, b, 1, epoch);
    await(c, b, 1, epoch);
  }
  if (tid == 0) st_sys(&mine->epoch[b], epoch);
}

// out[p][0 : bytes) = rank p's `in` (bytes <= 4 KiB, multiple of 16): ONE workgroup, protocol instance
// kMaxBlocks - 1 (kept apart from the row instances so grids of different sizes never share an epoch word).
__global__ __launch_bounds__(256) void allgather_small_kernel(CommDev c, const unsigned char* __restrict__ in,
                                                               unsigned char* __restrict__ out, int bytes) {
  const int b = kMaxBlocks - 1, tid = threadIdx.x;
  Flags* mine = reinterpret_cast<Flags*>(c.base[c.rank]);
  const uint32_t epoch = ld_sys(&mine->epoch[b]) + 1u;
  // the tail of the data region is reserved for this instance (see nvl_allreduce_create)
  const uint64_t off = c.red_off - 4096;
  if (tid * 16 < bytes)
    *reinterpret_cast<u32x4_t*>(c.base[c.rank] + off + tid * 16) = *reinterpret_cast<const u32x4_t*>(in + tid * 16);
  publish(c, b, 0, epoch);
  await(c, b, 0, epoch);
  for (int p = 0; p < c.world; ++p)
    if (tid * 16 < bytes)
      *reinterpret_cast<u32x4_t*>(out + (int64_t)p * bytes + tid * 16) =
          *reinterpret_cast<const u32x4_t*>(c.base[p] + off + tid * 16);
  // nobody may overwrite its slot for the next call before every peer has read it
  publish(c, b, 1, epoch);
  await(c, b, 1, epoch);
  if (tid == 0) st_sys(&mine->epoch[b], epoch);
}

// *out = OR over the W ranks of the latched spin-timeout word (every rank's flag region is mapped here, uncached): the
// serving path's view of nvl_allreduce_status — enqueue-only, so it can sit at the end of a captured decode step and travel
// to the host with the step's sampled ids. A timeout on ANY rank invalidates the step on every rank (the late rank's
// contribution was missing from the sums its peers used).
__global__ __launch_bounds__(64) void allreduce_status_kernel(CommDev c, uint32_t* __restrict__ out) {
  uint32_t e = 0;
  if ((int)threadIdx.x < c.world) e = ld_sys(&reinterpret_cast<const Flags*>(c.base[threadIdx.x])->error);
  const unsigned long long any = __ballot(e != 0);
  if (threadIdx.x == 0) *out = any ? 1u : 0u;
}

Comm* as_comm(void* h) { return reinterpret_cast<Comm*>(h); }

int check_rows(const Comm* cm, int64_t rows, int hidden, const char* who) {
  NVL_REQUIRE(cm && cm->connected, "%s: communicator not connected", who);
  NVL_REQUIRE(rows >= 0 && rows < (1ll << 31), "%s: bad rows=%lld", who, (long long)rows);
  NVL_REQUIRE(hidden > 0 && hidden % (8 * cm->dev.world) == 0 && hidden <= 256 * 8 * 4,
              "%s: hidden=%d must be a multiple of 8*world=%d and <= 8192", who, hidden, 8 * cm->dev.world);
  NVL_REQUIRE((size_t)rows * hidden * 2 <= cm->data_bytes, "%s: %lld x %d bf16 exceeds the %zu-byte comm buffer", who,
              (long long)rows, hidden, cm->data_bytes);
  return NVL_OK;
}

template <int EPI>
int launch_rows(Comm* cm, const void* in, void* out, void* residual, const void* weight, int64_t rows, int hidden,
                float eps, hipStream_t s) {
  const int grid = (int)(rows < kMaxBlocks - 1 ? rows : kMaxBlocks - 1);
  const bool oneshot = (size_t)rows * hidden * 2 <= 16 * 1024;   // latency-bound: skip the second exchange
  if (oneshot)
    hipLaunchKernelGGL((allreduce_rows_kernel<EPI, true>), dim3(grid), dim3(256), 0, s, cm->dev, (const bf16_t*)in,
                       (bf16_t*)out, (bf16_t*)residual, (const bf16_t*)weight, (int)rows, hidden, eps);
  else
    hipLaunchKernelGGL((allreduce_rows_kernel<EPI, false>), dim3(grid), dim3(256), 0, s, cm->dev, (const bf16_t*)in,
                       (bf16_t*)out, (bf16_t*)residual, (const bf16_t*)weight, (int)rows, hidden, eps);
  return nvl_check_launch("nvl_allreduce");
}

}  // namespace

extern "C" int nvl_allreduce_create(int rank, int world, int64_t max_bytes, void** comm_out) {
  NVL_REQUIRE(comm_out, "nvl_allreduce_create: null pointer");
  NVL_REQUIRE(world >= 2 && world <= kMaxWorld && rank >= 0 && rank < world, "nvl_allreduce_create: bad rank %d / world %d",
              rank, world);
  NVL_REQUIRE(max_bytes >= 4096 && max_bytes <= (1ll << 30), "nvl_allreduce_create: max_bytes=%lld out of range",
              (long long)max_bytes);
  Comm* cm = new Comm();
  memset(cm, 0, sizeof(*cm));
  const size_t data = (((size_t)max_bytes + 4095) / 4096) * 4096 + 4096;   // + the all-gather slot
  cm->data_bytes = data - 4096;
  cm->total_bytes = kFlagBytes + 2 * data;
  cm->dev.rank = rank;
  cm->dev.world = world;
  cm->dev.data_off = kFlagBytes;
  cm->dev.red_off = kFlagBytes + data;
  cm->dev.fences = 1;      // fenced unless the caller has validated the lean hand-off on ITS topology (tp.init_p2p)
  void* p = nullptr;
  const size_t total_bytes = cm->total_bytes;
  if (hipExtMallocWithFlags(&p, total_bytes, hipDeviceMallocUncached) != hipSuccess || !p) {
    (void)hipGetLastError();
    delete cm;
    nvl_set_error("nvl_allreduce_create: cannot allocate %zu B of uncached device memory", total_bytes);
    return NVL_ELAUNCH;
  }
  if (hipMemset(p, 0, cm->total_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(p);
    delete cm;
    nvl_set_error("nvl_allreduce_create: cannot clear the comm buffer");
    return NVL_ELAUNCH;
  }
  cm->dev.base[rank] = (unsigned char*)p;
  *comm_out = cm;
  return NVL_OK;
}

extern "C" int nvl_allreduce_uid(void* comm, void* uid_out) {
  Comm* cm = as_comm(comm);
  NVL_REQUIRE(cm && uid_out, "nvl_allreduce_uid: null pointer");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "uid is a 64-byte IPC memory handle");
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, cm->dev.base[cm->dev.rank]);
  if (e != hipSuccess) {
    nvl_set_error("nvl_allreduce_uid: hipIpcGetMemHandle: %s", hipGetErrorString(e));
    return NVL_ELAUNCH;
  }
  memcpy(uid_out, &h, 64);
  return NVL_OK;
}

extern "C" int nvl_allreduce_connect(void* comm, const void* uids) {
  Comm* cm = as_comm(comm);
  NVL_REQUIRE(cm && uids, "nvl_allreduce_connect: null pointer");
  NVL_REQUIRE(!cm->connected, "nvl_allreduce_connect: already connected");
  for (int p = 0; p < cm->dev.world; ++p) {
    if (p == cm->dev.rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, (const unsigned char*)uids + 64 * p, 64);
    void* ptr = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess || !ptr) {
      (void)hipGetLastError();
      nvl_set_error("nvl_allreduce_connect: hipIpcOpenMemHandle(rank %d): %s", p, hipGetErrorString(e));
      return NVL_ELAUNCH;
    }
    cm->dev.base[p] = (unsigned char*)ptr;
    cm->opened[p] = true;
  }
  cm->connected = true;
  return NVL_OK;
}

extern "C" int64_t nvl_allreduce_max_bytes(void* comm) {
  Comm* cm = as_comm(comm);
  return cm ? (int64_t)cm->data_bytes : 0;
}

extern "C" void* nvl_allreduce_buffer(void* comm) {
  Comm* cm = as_comm(comm);
  return cm ? (void*)(cm->dev.base[cm->dev.rank] + cm->dev.data_off) : nullptr;
}

extern "C" int nvl_allreduce_set_fences(void* comm, int on) {
  Comm* cm = as_comm(comm);
  NVL_REQUIRE(cm, "nvl_allreduce_set_fences: null pointer");
  cm->dev.fences = on ? 1 : 0;
  return NVL_OK;
}

extern "C" int nvl_allreduce_run(void* comm, const void* in, void* out, int64_t rows, int hidden, void* stream) {
  Comm* cm = as_comm(comm);
  NVL_REQUIRE(in && out, "nvl_allreduce_run: null pointer");
  NVL_REQUIRE(((uintptr_t)in | (uintptr_t)out) % 16 == 0, "nvl_allreduce_run: pointers must be 16-byte aligned");
  if (int rc = check_rows(cm, rows, hidden, "nvl_allreduce_run")) return rc;
  if (rows == 0) return NVL_OK;
  return launch_rows<EPI_STORE>(cm, in, out, nullptr, nullptr, rows, hidden, 0.f, (hipStream_t)stream);
}

extern "C" int nvl_allreduce_add_rmsnorm(void* comm, const void* x_partial, void* residual, const void* weight, void* y,
                                         int64_t rows, int hidden, float eps, void* stream) {
  Comm* cm = as_comm(comm);
  NVL_REQUIRE(x_partial && residual && weight && y, "nvl_allreduce_add_rmsnorm: null pointer");
  NVL_REQUIRE(((uintptr_t)x_partial | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)y) % 16 == 0,
              "nvl_allreduce_add_rmsnorm: pointers must be 16-byte aligned");
  if (int rc = check_rows(cm, rows, hidden, "nvl_allreduce_add_rmsnorm")) return rc;
  if (rows == 0) return NVL_OK;
  return launch_rows<EPI_ADD_RMSNORM>(cm, x_partial, y, residual, weight, rows, hidden, eps, (hipStream_t)stream);
}

extern "C" int nvl_allreduce_gather(void* comm, const void* in, void* out, int64_t bytes_per_rank, void* stream) {
  Comm* cm = as_comm(comm);
  NVL_REQUIRE(cm && cm->connected, "nvl_allreduce_gather: communicator not connected");
  NVL_REQUIRE(in && out, "nvl_allreduce_gather: null pointer");
  NVL_REQUIRE(bytes_per_rank >= 0 && bytes_per_rank <= 4096 && bytes_per_rank % 16 == 0,
              "nvl_allreduce_gather: bytes_per_rank=%lld must be a multiple of 16 and <= 4096", (long long)bytes_per_rank);
  NVL_REQUIRE(((uintptr_t)in | (uintptr_t)out) % 16 == 0, "nvl_allreduce_gather: pointers must be 16-byte aligned");
  if (bytes_per_rank == 0) return NVL_OK;
  hipLaunchKernelGGL(allgather_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, cm->dev,
                     (const unsigned char*)in, (unsigned char*)out, (int)bytes_per_rank);
  return nvl_check_launch("nvl_allreduce_gather");
}

extern "C" int nvl_allreduce_status(void* comm) {
  Comm* cm = as_comm(comm);
  NVL_REQUIRE(cm, "nvl_allreduce_status: null pointer");
  uint32_t err = 0;
  const Flags* f = reinterpret_cast<const Flags*>(cm->dev.base[cm->dev.rank]);
  if (hipMemcpy(&err, &f->error, sizeof(err), hipMemcpyDeviceToHost) != hipSuccess) {
    nvl_set_error("nvl_allreduce_status: cannot read the status word");
    return NVL_ELAUNCH;
  }
  if (err) {
    nvl_set_error("nvl_allreduce: a peer did not arrive within the spin limit (result of that call is invalid)");
    return NVL_ELAUNCH;
  }
  return NVL_OK;
}

extern "C" int nvl_allreduce_status_async(void* comm, void* status_out, void* stream) {
  Comm* cm = as_comm(comm);
  NVL_REQUIRE(cm && cm->connected, "nvl_allreduce_status_async: communicator not connected");
  NVL_REQUIRE(status_out && (uintptr_t)status_out % 4 == 0, "nvl_allreduce_status_async: status_out must be a 4-byte aligned device pointer");
  hipLaunchKernelGGL(allreduce_status_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cm->dev, (uint32_t*)status_out);
  return nvl_check_launch("nvl_allreduce_status_async");
}

extern "C" int nvl_allreduce_destroy(void* comm) {
  Comm* cm = as_comm(comm);
  if (!cm) return NVL_OK;
  for (int p = 0; p < cm->dev.world; ++p)
    if (cm->opened[p]) (void)hipIpcCloseMemHandle(cm->dev.base[p]);
  if (cm->dev.base[cm->dev.rank]) (void)hipFree(cm->dev.base[cm->dev.rank]);
  delete cm;
  return NVL_OK;
}
