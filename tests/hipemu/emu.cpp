// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/hipemu/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

#include <vector>

emu_uint3 threadIdx, blockIdx, blockDim, gridDim;
unsigned char* emu::dyn_shared = nullptr;

namespace {

constexpr size_t kStackSize = 256 * 1024;

enum State { READY, WAIT_WAVE, WAIT_BLOCK, DONE };

struct Fiber {
  void* sp = nullptr;
  unsigned char* stack = nullptr;
  State state = READY;
  unsigned tid = 0;
  uint64_t contrib = 0;   // value offered to the pending wave collective
  unsigned seq = 0;       // number of collectives this lane has entered (per wave)
  unsigned bseq = 0;      // number of block barriers entered
};

struct Wave {
  uint64_t exch[2][64];
  uint64_t live_at[2];
  unsigned gen = 0;  // completed collectives
};

std::vector<Fiber> g_fibers;
std::vector<Wave> g_waves;
unsigned g_block_gen = 0;
Fiber* g_cur = nullptr;
void* g_sched_sp = nullptr;
const std::function<void()>* g_body = nullptr;

extern "C" void emu_ctx_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_ctx_switch, .-emu_ctx_switch
)");

void yield_to_scheduler() { emu_ctx_switch(&g_cur->sp, g_sched_sp); }

void fiber_entry() {
  (*g_body)();
  g_cur->state = DONE;
  yield_to_scheduler();
  abort();  // never resumed
}

void init_fiber(Fiber& f, unsigned tid) {
  if (!f.stack) f.stack = (unsigned char*)aligned_alloc(64, kStackSize);
  f.tid = tid;
  f.state = READY;
  f.seq = f.bseq = 0;
  uintptr_t top = ((uintptr_t)f.stack + kStackSize) & ~(uintptr_t)15;
  uint64_t* sp = (uint64_t*)top;
  *--sp = 0;                        // fake return address of fiber_entry's "caller"
  *--sp = (uint64_t)&fiber_entry;   // consumed by `ret` in emu_ctx_switch
  for (int i = 0; i < 6; i++) *--sp = 0;  // rbp rbx r12..r15
  f.sp = sp;
}

[[noreturn]] void die(const char* msg) {
  fprintf(stderr, "hipemu: %s (block %u)\n", msg, blockIdx.x);
  abort();
}

}  // namespace

unsigned emu::lane_id() { return g_cur->tid & 63; }

const uint64_t* emu::wave_exchange(uint64_t my_value, uint64_t* live_mask) {
  Fiber* me = g_cur;
  Wave& w = g_waves[me->tid / 64];
  unsigned my_gen = me->seq++;
  if (my_gen != w.gen) die("wave collective out of order (divergent cross-lane op?)");
  me->contrib = my_value;
  me->state = WAIT_WAVE;
  yield_to_scheduler();  // resumed once the scheduler has completed collective my_gen
  *live_mask = w.live_at[my_gen & 1];
  return w.exch[my_gen & 1];
}

void emu::block_barrier() {
  Fiber* me = g_cur;
  me->bseq++;
  me->state = WAIT_BLOCK;
  yield_to_scheduler();
}

void emu::launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  if (grid.y != 1 || grid.z != 1 || block.y != 1 || block.z != 1) die("emu supports 1-D launches only");
  unsigned nthreads = block.x;
  static unsigned char* dyn = nullptr;
  static size_t dyn_cap = 0;
  if (shmem > dyn_cap) {
    free(dyn);
    dyn = (unsigned char*)aligned_alloc(64, (shmem + 63) & ~(size_t)63);
    dyn_cap = shmem;
  }
  emu::dyn_shared = dyn;
  if (g_fibers.size() < nthreads) g_fibers.resize(nthreads);
  unsigned nwaves = (nthreads + 63) / 64;
  g_waves.resize(nwaves);
  gridDim = {grid.x, 1, 1};
  blockDim = {nthreads, 1, 1};
  g_body = &body;

  for (unsigned b = 0; b < grid.x; b++) {
    blockIdx = {b, 0, 0};
    for (unsigned t = 0; t < nthreads; t++) init_fiber(g_fibers[t], t);
    for (auto& w : g_waves) w.gen = 0;
    g_block_gen = 0;
    unsigned done = 0;
    while (done < nthreads) {
      bool progressed = false;
      // run every READY fiber until it blocks or finishes
      for (unsigned t = 0; t < nthreads; t++) {
        Fiber& f = g_fibers[t];
        if (f.state != READY) continue;
        g_cur = &f;
        threadIdx = {t, 0, 0};
        emu_ctx_switch(&g_sched_sp, f.sp);
        progressed = true;
        if (f.state == DONE) done++;
      }
      // complete wave collectives whose live lanes have all arrived
      for (unsigned wv = 0; wv < nwaves; wv++) {
        Wave& w = g_waves[wv];
        unsigned lo = wv * 64, hi = lo + 64 < nthreads ? lo + 64 : nthreads;
        unsigned waiting = 0, live = 0;
        for (unsigned t = lo; t < hi; t++) {
          if (g_fibers[t].state == DONE) continue;
          live++;
          if (g_fibers[t].state == WAIT_WAVE) waiting++;
        }
        if (live && waiting == live) {
          unsigned par = w.gen & 1;
          uint64_t mask = 0;
          for (unsigned i = 0; i < 64; i++) w.exch[par][i] = 0;
          for (unsigned t = lo; t < hi; t++)
            if (g_fibers[t].state == WAIT_WAVE) {
              if (g_fibers[t].seq != w.gen + 1) die("wave collective sequence mismatch");
              w.exch[par][t - lo] = g_fibers[t].contrib;
              mask |= 1ull << (t - lo);
              g_fibers[t].state = READY;
            }
          w.live_at[par] = mask;
          w.gen++;
          progressed = true;
        }
      }
      // block barrier
      {
        unsigned waiting = 0, live = 0;
        for (unsigned t = 0; t < nthreads; t++) {
          if (g_fibers[t].state == DONE) continue;
          live++;
          if (g_fibers[t].state == WAIT_BLOCK) waiting++;
        }
        if (live && waiting == live) {
          for (unsigned t = 0; t < nthreads; t++)
            if (g_fibers[t].state == WAIT_BLOCK) {
              if (g_fibers[t].bseq != g_block_gen + 1) die("__syncthreads count mismatch");
              g_fibers[t].state = READY;
            }
          g_block_gen++;
          progressed = true;
        }
      }
      if (!progressed) die("deadlock: lanes wait at different rendezvous points");
    }
  }
  g_body = nullptr;
}
