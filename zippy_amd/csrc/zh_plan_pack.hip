// Device-resident plans across GPUs: a plan's results packed back to back, and streams scattered into a plan's
// source slots -- the device side of "the trivial scatter / gather of whole buffers" (zippy.nim:11-18: a buffer is
// compressed or uncompressed by itself, so whole buffers are all that ever travels).  Output slots are sparse (a
// slot has the worst-case size, a stream fills 40 % of it); what goes over the wire is the streams alone:
//
//   zh_plan_pack    slots + the plan's device-side lengths -> streams back to back + n + 1 device offsets
//   zh_plan_unpack  streams back to back + offsets -> an uncompress plan's source slots + its device-side lengths
//
// Both are two launches on the context's stream and never touch the host: the lengths are the ones zh_plan_run left
// on the device (zh_plan_device_lens), so a pack can follow a run without a synchronisation in between.
#include "zh_host.h"

namespace {

struct __attribute__((packed)) PackVec16 {
  uint32_t w[4];
};

// offsets[0] = 0, offsets[i + 1] = offsets[i] + (the stream's length: out_len[i] where status[i] is ZH_OK, else 0)
// -- one workgroup, 1024 streams a turn.
__global__ __launch_bounds__(1024) void zh_pack_offsets_kernel(uint32_t n, const uint64_t* __restrict__ out_len,
                                                               const int32_t* __restrict__ status,
                                                               uint64_t* __restrict__ offsets) {
  __shared__ uint64_t s_wave[16];
  __shared__ uint64_t s_carry;
  const uint32_t tid = threadIdx.x;
  const unsigned lane = zh_lane();
  if (tid == 0) {
    s_carry = 0;
    offsets[0] = 0;
  }
  __syncthreads();
  for (uint32_t base = 0; base < n; base += 1024u) {
    const uint32_t i = base + tid;
    const uint64_t len = i < n && status[i] == ZH_OK ? out_len[i] : 0ull;
    uint64_t incl = len;  // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t t = __shfl_up(incl, o, 64);
      if (lane >= (unsigned)o) incl += t;
    }
    if (lane == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    uint64_t before = s_carry;
    for (uint32_t w = 0; w < (tid >> 6); w++) before += s_wave[w];
    if (i < n) offsets[i + 1] = before + incl;
    __syncthreads();
    if (tid == 1023) s_carry = before + incl;
    __syncthreads();
  }
}

// lens[i] = min(offsets[i + 1] - offsets[i], the slot's capacity): what zh_plan_unpack leaves as the plan's
// device-side source lengths
__global__ __launch_bounds__(256) void zh_unpack_lens_kernel(uint32_t n, const ZhBufDesc* __restrict__ bufs,
                                                             const uint64_t* __restrict__ offsets,
                                                             uint64_t* __restrict__ lens) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint64_t len = offsets[i + 1] - offsets[i];
  lens[i] = len < bufs[i].src_len ? len : bufs[i].src_len;
}

// One piece (`piece` bytes of a slot at most) of one buffer a workgroup: kToSlots false: slot -> packed (the plan's
// output slots, dst_off), true: packed -> slot (the plan's source slots, src_off, at most src_len bytes).  Sixteen
// bytes a thread and turn, the stores aligned (the loads need not be on gfx950), bytes at both ends.
template <bool kToSlots>
__global__ __launch_bounds__(256) void zh_pack_copy_kernel(const uint8_t* __restrict__ from, uint8_t* __restrict__ to,
                                                           const ZhBufDesc* __restrict__ bufs,
                                                           const uint64_t* __restrict__ offsets, uint32_t pieces,
                                                           uint64_t piece, uint64_t packed_cap) {
  const uint32_t i = blockIdx.x / pieces, k = blockIdx.x % pieces;
  const ZhBufDesc bd = bufs[i];
  const uint64_t off = offsets[i];
  uint64_t len = offsets[i + 1] - off;
  if (kToSlots) {
    if (len > bd.src_len) len = bd.src_len;
  } else {
    if (off >= packed_cap) return;  // (nothing is written past the packed buffer; the caller compares offsets[n] with it)
    if (len > packed_cap - off) len = packed_cap - off;
  }
  const uint64_t lo = (uint64_t)k * piece;
  if (lo >= len) return;
  const uint64_t cnt = len - lo < piece ? len - lo : piece;
  const uint8_t* s = from + (kToSlots ? off : bd.dst_off) + lo;
  uint8_t* d = to + (kToSlots ? bd.src_off : off) + lo;
  uint64_t head = (16u - (uint32_t)((uintptr_t)d & 15u)) & 15u;
  if (head > cnt) head = cnt;
  if (threadIdx.x < head) d[threadIdx.x] = s[threadIdx.x];
  const uint64_t nv = (cnt - head) >> 4;
  const PackVec16* sv = reinterpret_cast<const PackVec16*>(s + head);
  uint4* dv = reinterpret_cast<uint4*>(d + head);
  for (uint64_t j = threadIdx.x; j < nv; j += 256u) {
    const PackVec16 v = sv[j];
    dv[j] = make_uint4(v.w[0], v.w[1], v.w[2], v.w[3]);
  }
  for (uint64_t j = head + (nv << 4) + threadIdx.x; j < cnt; j += 256u) d[j] = s[j];
}

// pieces of 64 KiB, or larger where a slot would need more than 4096 of them
void piece_geometry(uint64_t max_bytes, uint32_t* pieces, uint64_t* piece) {
  uint64_t p = 65536;
  while ((max_bytes + p - 1) / p > 4096u) p <<= 1;
  *piece = p;
  *pieces = (uint32_t)std::max<uint64_t>(1, (max_bytes + p - 1) / p);
}

}  // namespace

extern "C" int zh_plan_pack(zh_plan* plan, const void* d_slots, void* d_packed, uint64_t packed_cap, uint64_t* d_offsets) {
  if (!plan) return ZH_ERR_ARGUMENT;
  zh_ctx* ctx = plan->ctx;
  if (!plan->n) {
    if (!d_offsets) return ZH_ERR_ARGUMENT;
    ZH_HIP(ctx, hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), ctx->stream));
    return ZH_OK;
  }
  if (!d_slots || !d_packed || !d_offsets || plan->indexed) return ZH_ERR_ARGUMENT;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  uint32_t pieces;
  uint64_t piece;
  piece_geometry(plan->dst_max_cap, &pieces, &piece);
  if ((uint64_t)plan->n * pieces > 0x7fffffffull) return ZH_ERR_ARGUMENT;  // (before anything is launched or written)
  hipLaunchKernelGGL(zh_pack_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, (uint32_t)plan->n, plan->out_len,
                     plan->status, d_offsets);
  hipLaunchKernelGGL((zh_pack_copy_kernel<false>), dim3((uint32_t)plan->n * pieces), dim3(256), 0, ctx->stream,
                     static_cast<const uint8_t*>(d_slots), static_cast<uint8_t*>(d_packed), plan->d_bufs, d_offsets, pieces,
                     piece, packed_cap);
  ZH_HIP(ctx, hipGetLastError());
  return ZH_OK;
}

extern "C" int zh_plan_unpack(zh_plan* plan, const void* d_packed, const uint64_t* d_offsets, void* d_slots) {
  if (!plan || plan->is_compress || plan->indexed) return ZH_ERR_ARGUMENT;
  zh_ctx* ctx = plan->ctx;
  if (!plan->n) return ZH_OK;
  if (!d_packed || !d_offsets || !d_slots) return ZH_ERR_ARGUMENT;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  uint32_t pieces;
  uint64_t piece;
  piece_geometry(plan->src_max_len, &pieces, &piece);
  if ((uint64_t)plan->n * pieces > 0x7fffffffull) return ZH_ERR_ARGUMENT;  // (before anything is launched or written)
  if (!plan->unpack_lens) {
    void* p = nullptr;
    ZH_HIP(ctx, ctx_malloc(ctx, &p, plan->n * sizeof(uint64_t)));
    plan->unpack_lens = static_cast<uint64_t*>(p);
  }
  hipLaunchKernelGGL(zh_unpack_lens_kernel, dim3(((uint32_t)plan->n + 255u) / 256u), dim3(256), 0, ctx->stream,
                     (uint32_t)plan->n, plan->d_bufs, d_offsets, plan->unpack_lens);
  hipLaunchKernelGGL((zh_pack_copy_kernel<true>), dim3((uint32_t)plan->n * pieces), dim3(256), 0, ctx->stream,
                     static_cast<const uint8_t*>(d_packed), static_cast<uint8_t*>(d_slots), plan->d_bufs, d_offsets, pieces,
                     piece, ~0ull);
  ZH_HIP(ctx, hipGetLastError());
  return zh_plan_set_src_lens_device(plan, plan->unpack_lens);
}
