// lrnde_comm.hpp — the exchange layer of a batch-sharded handle (the reference has no collective:
// §2 rows 16-17).  Every collective of liblrnde is a SUM all-reduce and goes through comm_allreduce():
//   * RCCL (`ncclAllReduce` on the handle's stream) when the handle joined a communicator with lrnde_comm_init —
//     one process per GPU, the product path;
//   * the in-process LOCAL communicator (lrnde_local_comm_*, include/lrnde_hooks.h) when several handles of one
//     process — driven by one host thread each, all on ONE device (lrnde_comm_init_local refuses a second device: the sum kernel reads peers' buffers directly and no peer access is set up) — were joined with
//     lrnde_comm_init_local.  It performs the same reduction with stream-ordered kernels and events, so that the
//     nranks > 1 code of the library (offsets into the partial-sum vectors, receive buffers, the sharded adjoint's
//     norm and parameter-cotangent sums) runs on a one-GPU box, where RCCL refuses two ranks on one device.
// Included by lrnde_kernels.hip at file scope (before the anonymous-namespace host helpers).
#pragma once

#include <chrono>
#include <condition_variable>
#include <mutex>

constexpr int LRNDE_LC_MAXR = 16;

struct lrnde_local_comm {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long gen = 0;
  bool broken = false;  // a rank timed out or failed: every later barrier fails at once instead of hanging
  // per-rank slots of the collective in flight
  const void* send[LRNDE_LC_MAXR] = {};
  void* tmp[LRNDE_LC_MAXR] = {};
  size_t tmp_bytes[LRNDE_LC_MAXR] = {};
  hipEvent_t ready[LRNDE_LC_MAXR] = {};  // rank's send buffer is complete (recorded on its stream)
  hipEvent_t done[LRNDE_LC_MAXR] = {};   // rank has finished READING every peer's send buffer
  int device[LRNDE_LC_MAXR] = {};
  bool joined[LRNDE_LC_MAXR] = {};
};

namespace {

// host barrier over the n rank threads; false on timeout / broken communicator (never hangs a GPU box)
inline bool lc_barrier(lrnde_local_comm* lc) {
  std::unique_lock<std::mutex> lk(lc->mu);
  if (lc->broken) return false;
  const unsigned long g = lc->gen;
  if (++lc->arrived == lc->n) {
    lc->arrived = 0;
    ++lc->gen;
    lc->cv.notify_all();
    return true;
  }
  const bool ok = lc->cv.wait_for(lk, std::chrono::seconds(60), [&] { return lc->gen != g || lc->broken; });
  if (!ok || lc->broken) {
    lc->broken = true;
    lc->cv.notify_all();
    return false;
  }
  return true;
}

struct LcPtrs { const void* p[LRNDE_LC_MAXR]; };

// out[i] = p[0][i] + p[1][i] + ... in rank order (what a sum all-reduce returns; with the zero-padded vectors the
// library exchanges every element has one non-zero term, so the order does not matter there)
template <class T> __global__ void k_lc_sum(LcPtrs s, int n, size_t count, T* out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    T acc = reinterpret_cast<const T*>(s.p[0])[i];
    for (int r = 1; r < n; ++r) acc = acc + reinterpret_cast<const T*>(s.p[r])[i];
    out[i] = acc;
  }
}

// sum all-reduce over the local communicator; send may equal recv (in place).  Returns 0 or a LRNDE status.
inline int lc_allreduce(lrnde_local_comm* lc, int rank, hipStream_t stream, const void* send, void* recv, size_t count,
                        bool is_double) {
  const size_t bytes = count * (is_double ? sizeof(double) : sizeof(float));
  if (lc->tmp_bytes[rank] < bytes) {
    if (lc->tmp[rank]) (void)hipFree(lc->tmp[rank]);
    lc->tmp[rank] = nullptr; lc->tmp_bytes[rank] = 0;
    if (hipMalloc(&lc->tmp[rank], bytes) != hipSuccess) return LRNDE_HIP_ERROR;
    lc->tmp_bytes[rank] = bytes;
  }
  lc->send[rank] = send;
  if (hipEventRecord(lc->ready[rank], stream) != hipSuccess) return LRNDE_HIP_ERROR;
  if (!lc_barrier(lc)) return LRNDE_NCCL_ERROR;
  LcPtrs s;
  for (int r = 0; r < lc->n; ++r) {
    s.p[r] = lc->send[r];
    if (r != rank && hipStreamWaitEvent(stream, lc->ready[r], 0) != hipSuccess) return LRNDE_HIP_ERROR;
  }
  int nb = (int)((count + 255) / 256); if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
  if (is_double) hipLaunchKernelGGL(k_lc_sum<double>, dim3(nb), dim3(256), 0, stream, s, lc->n, count, (double*)lc->tmp[rank]);
  else hipLaunchKernelGGL(k_lc_sum<float>, dim3(nb), dim3(256), 0, stream, s, lc->n, count, (float*)lc->tmp[rank]);
  if (hipGetLastError() != hipSuccess) return LRNDE_HIP_ERROR;
  if (hipEventRecord(lc->done[rank], stream) != hipSuccess) return LRNDE_HIP_ERROR;
  if (!lc_barrier(lc)) return LRNDE_NCCL_ERROR;
  // nobody may overwrite a send buffer (this rank's recv may BE its send buffer) before every peer has read it
  for (int r = 0; r < lc->n; ++r)
    if (r != rank && hipStreamWaitEvent(stream, lc->done[r], 0) != hipSuccess) return LRNDE_HIP_ERROR;
  if (hipMemcpyAsync(recv, lc->tmp[rank], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return LRNDE_HIP_ERROR;
  return LRNDE_OK;
}

}  // namespace
