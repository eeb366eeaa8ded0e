// dexr_hostctx.hpp -- persistent staging for the HOST-pointer entry points of libdexr.so.
//
// The reference's callers hand over one frame at a time (SeqRetargeting.retarget, 621 calls in
// /root/reference/example/profiling/profile_online_retargeting.py:18-36): for that regime the cost of a call is the
// host path around the kernel, not the kernel.  A context uses
//   * the library's private non-blocking stream of the current device (no hipDeviceSynchronize: other streams of the
//     process are never stalled; ONE stream per device for all handles -- HIP multiplexes streams onto a handful of
//     hardware queues, and a stream per model handle would soon share a queue with the caller's own streams),
//   * its own grow-only pinned host buffer and grow-only device buffer,
// and a call packs every array into ONE contiguous block laid out  [ inputs | in-out | outputs ]  so that it costs one
// host-to-device copy of [inputs | in-out], the launches, and one device-to-host copy of [in-out | outputs], all on the
// private stream, then one hipStreamSynchronize.  No hipMalloc / hipFree on the call path once the buffers have grown.
// SMALL calls (<= ZC_MAX bytes, the one-frame-per-call regime) whose in-out arrays all start as zeros skip the host-to-device
// copy altogether: the kernels read the inputs straight from the pinned block (it is device-accessible; a few hundred bytes
// over PCIe cost less than a copy command and its completion signal), and the zeroed in-out region of the device block is
// prepared by a memset queued behind the PREVIOUS call's copy-back, off the critical path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>

namespace dexr {

// the library's private stream of the current device (created on first use, never destroyed)
inline hipError_t host_stream(hipStream_t* out) {
  static std::mutex mu;
  static hipStream_t streams[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  if (!streams[dev]) {
    e = hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking);
    if (e != hipSuccess) return e;
  }
  *out = streams[dev];
  return hipSuccess;
}

struct HostCtx {
  std::mutex mu;  // host-pointer calls through one context are serialised (the staging block is shared)
  hipStream_t st = nullptr;  // = host_stream() of the device the context was first used on (not owned)
  unsigned char* pin = nullptr;
  unsigned char* pin_dev = nullptr;  // the pinned block's address as the device sees it (zero-copy reads of small calls)
  size_t pin_bytes = 0;
  size_t zero_lo = 0, zero_hi = 0;   // [zero_lo, zero_hi) of `dev` holds zeros (memset queued on `st` after a small call)
  unsigned char* dev = nullptr;
  size_t dev_bytes = 0;
  // above this size the block is copied piecewise from / to the caller's (pageable) arrays instead of through the pinned
  // mirror: packing costs a host memcpy per array, which pays for small calls (one copy each way instead of six) and
  // loses for large ones -- measured, Allegro vector through dexr_retarget_kp: 65 536 frames 1.20 ms packed vs 0.6 ms
  // piecewise (tools/host_path_rate.py)
  static constexpr size_t PIN_MAX = (size_t)1 << 20;
  static constexpr size_t ZC_MAX = (size_t)16 << 10;  // zero-copy inputs up to this block size

  ~HostCtx() { release(); }
  void release() {
    if (dev) (void)hipFree(dev);
    if (pin) (void)hipHostFree(pin);
    dev = pin = pin_dev = nullptr;
    st = nullptr;
    dev_bytes = pin_bytes = 0;
    zero_lo = zero_hi = 0;
  }
  hipError_t ensure(size_t bytes) {
    hipError_t e = hipSuccess;
    if (!st) {
      e = host_stream(&st);
      if (e != hipSuccess) return e;
    }
    if (dev_bytes < bytes) {
      if (dev) {
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
        (void)hipFree(dev);
        dev = nullptr;
        dev_bytes = 0;
        zero_lo = zero_hi = 0;
      }
      size_t want = dev_bytes ? dev_bytes : 4096;
      while (want < bytes) want *= 2;
      e = hipMalloc((void**)&dev, want);
      if (e != hipSuccess) return e;
      dev_bytes = want;
    }
    if (pin_bytes < bytes && bytes <= PIN_MAX) {
      if (pin) {
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
        (void)hipHostFree(pin);
        pin = pin_dev = nullptr;
        pin_bytes = 0;
      }
      size_t want = 4096;
      while (want < bytes) want *= 2;
      e = hipHostMalloc((void**)&pin, want, hipHostMallocDefault);
      if (e != hipSuccess) return e;
      pin_bytes = want;
      void* dp = nullptr;
      pin_dev = (hipHostGetDevicePointer(&dp, pin, 0) == hipSuccess) ? static_cast<unsigned char*>(dp) : nullptr;
    }
    return hipSuccess;
  }
};

// One packed block.  Segments are declared in the order inputs, in-outs, outputs; a NULL host pointer on an input /
// in-out segment means "zeros", on an output segment "not wanted" (the device region still exists).
class Staging {
 public:
  enum Dir { IN = 0, INOUT = 1, OUT = 2 };
  static constexpr int MAXSEG = 16;

  // returns the segment's index; offsets are 16-byte aligned
  int add(Dir dir, const void* src, void* dst, size_t bytes) {
    Seg& s = seg_[n_];
    s.dir = dir;
    s.src = src;
    s.dst = dst;
    s.bytes = bytes;
    s.off = total_;
    total_ += (bytes + 15) & ~(size_t)15;
    if (dir == IN) in_end_ = total_;
    if (dir != OUT) h2d_end_ = total_;
    if (dir == INOUT && src && bytes) inout_src_ = true;
    if (dir != IN && dst && bytes) d2h_end_ = total_;  // trailing scratch segments (no destination) are not copied back
    return n_++;
  }
  size_t total() const { return total_ ? total_ : 16; }

  hipError_t upload(HostCtx& c) {
    hipError_t e = c.ensure(total());
    if (e != hipSuccess) return e;
    pinned_ = c.pin && c.pin_bytes >= total();
    const bool zeros_ready = c.zero_lo <= in_end_ && h2d_end_ <= c.zero_hi;
    c.zero_lo = c.zero_hi = 0;  // whatever runs now dirties the device block
    zc_ = pinned_ && c.pin_dev && total() <= HostCtx::ZC_MAX && !inout_src_;
    if (zc_) {  // inputs stay in the pinned block; the in-out region of the device block must hold zeros
      for (int i = 0; i < n_; ++i) {
        const Seg& s = seg_[i];
        if (s.dir != IN || !s.bytes) continue;
        if (s.src) std::memcpy(c.pin + s.off, s.src, s.bytes);
        else std::memset(c.pin + s.off, 0, s.bytes);
      }
      if (!zeros_ready && h2d_end_ > in_end_) e = hipMemsetAsync(c.dev + in_end_, 0, h2d_end_ - in_end_, c.st);
      return e;
    }
    if (pinned_) {
      for (int i = 0; i < n_; ++i) {
        const Seg& s = seg_[i];
        if (s.dir == OUT || !s.bytes) continue;
        if (s.src) std::memcpy(c.pin + s.off, s.src, s.bytes);
        else std::memset(c.pin + s.off, 0, s.bytes);
      }
      if (h2d_end_) e = hipMemcpyAsync(c.dev, c.pin, h2d_end_, hipMemcpyHostToDevice, c.st);
      return e;
    }
    for (int i = 0; i < n_ && e == hipSuccess; ++i) {  // large batches: straight from the caller's arrays
      const Seg& s = seg_[i];
      if (s.dir == OUT || !s.bytes) continue;
      if (s.src) e = hipMemcpyAsync(c.dev + s.off, s.src, s.bytes, hipMemcpyHostToDevice, c.st);
      else e = hipMemsetAsync(c.dev + s.off, 0, s.bytes, c.st);
    }
    return e;
  }

  // copies back, waits for the private stream, scatters into the caller's arrays
  hipError_t download(HostCtx& c) {
    hipError_t e = hipSuccess;
    if (pinned_) {
      if (d2h_end_ > in_end_) e = hipMemcpyAsync(c.pin + in_end_, c.dev + in_end_, d2h_end_ - in_end_, hipMemcpyDeviceToHost, c.st);
      if (e == hipSuccess) e = hipStreamSynchronize(c.st);
      if (e != hipSuccess) return e;
      for (int i = 0; i < n_; ++i) {
        const Seg& s = seg_[i];
        if (s.dir != IN && s.dst && s.bytes) std::memcpy(s.dst, c.pin + s.off, s.bytes);
      }
      if (zc_ && h2d_end_ > in_end_) {  // the next small call finds its in-out region zeroed (not waited for here)
        if (hipMemsetAsync(c.dev + in_end_, 0, h2d_end_ - in_end_, c.st) == hipSuccess) {
          c.zero_lo = in_end_;
          c.zero_hi = h2d_end_;
        }
      }
      return hipSuccess;
    }
    for (int i = 0; i < n_ && e == hipSuccess; ++i) {
      const Seg& s = seg_[i];
      if (s.dir != IN && s.dst && s.bytes) e = hipMemcpyAsync(s.dst, c.dev + s.off, s.bytes, hipMemcpyDeviceToHost, c.st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c.st);
    return e;
  }

  template <typename T> T* dev(const HostCtx& c, int i) const {
    return reinterpret_cast<T*>(((zc_ && seg_[i].dir == IN) ? c.pin_dev : c.dev) + seg_[i].off);
  }

 private:
  struct Seg {
    Dir dir;
    const void* src;
    void* dst;
    size_t bytes, off;
  };
  Seg seg_[MAXSEG];
  int n_ = 0;
  size_t total_ = 0, in_end_ = 0, h2d_end_ = 0, d2h_end_ = 0;
  bool pinned_ = false, zc_ = false, inout_src_ = false;
};

}  // namespace dexr
