// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emu/include/hip/hip_runtime.h.
// One ucontext fiber per HIP thread; blocks run sequentially; __syncthreads() and the
// wave-level exchange used by the MFMA / shuffle emulation are cooperative barriers.
#include <hip/hip_runtime.h>
#include <ucontext.h>
#include <vector>
#include <sys/mman.h>

emu_uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
unsigned char* emu_dyn_smem = nullptr;
static size_t emu_dyn_cap = 0;
void emu_set_dyn_smem(size_t bytes) {
    if (bytes > 160 * 1024) { fprintf(stderr, "emu: dynamic LDS request %zu > 160 KiB\n", bytes); abort(); }
    if (bytes > emu_dyn_cap) {
        free(emu_dyn_smem);
        emu_dyn_cap = bytes + 4096;
        emu_dyn_smem = (unsigned char*)aligned_alloc(64, (emu_dyn_cap + 63) & ~(size_t)63);
    }
}

namespace {
constexpr size_t kStack = 192 * 1024;
constexpr int kSlotBytes = 64;

struct Fiber {
    ucontext_t ctx;
    emu_uint3 tid;
    int linear;
    bool done;
};

struct Block {
    std::vector<Fiber> fibers;
    int nthreads = 0;
    int cur = 0;
    // block barrier
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    // per-wave exchange
    std::vector<int> wv_arrived;
    std::vector<unsigned> wv_gen;
    std::vector<unsigned char> wv_buf;   // [nwaves][64][kSlotBytes]
};

Block* g_blk = nullptr;
ucontext_t g_sched;
const std::function<void()>* g_body = nullptr;
char* g_stacks = nullptr;
size_t g_stacks_cap = 0;

void fiber_entry() {
    (*g_body)();
    g_blk->fibers[g_blk->cur].done = true;
    swapcontext(&g_blk->fibers[g_blk->cur].ctx, &g_sched);
}

inline void yield_to_sched() {
    Fiber& f = g_blk->fibers[g_blk->cur];
    swapcontext(&f.ctx, &g_sched);
    threadIdx = f.tid;   // restored after we are resumed
}

inline int wave_size(int wave) {
    const int rem = g_blk->nthreads - wave * 64;
    return rem >= 64 ? 64 : rem;
}

void wave_sync() {
    Block* b = g_blk;
    const int wave = b->fibers[b->cur].linear >> 6;
    const unsigned gen = b->wv_gen[wave];
    if (++b->wv_arrived[wave] == wave_size(wave)) {
        b->wv_arrived[wave] = 0;
        b->wv_gen[wave]++;
        return;
    }
    while (b->wv_gen[wave] == gen) yield_to_sched();
}
}  // namespace

int emu_lane_id() { return g_blk->fibers[g_blk->cur].linear & 63; }

void emu_wave_sync() { wave_sync(); }

void emu_syncthreads() {
    Block* b = g_blk;
    const unsigned gen = b->bar_gen;
    if (++b->bar_arrived == b->nthreads) {
        b->bar_arrived = 0;
        b->bar_gen++;
        return;
    }
    while (b->bar_gen == gen) yield_to_sched();
}

void emu_wave_exchange_begin(const void* src, size_t bytes) {
    if (bytes > (size_t)kSlotBytes) { fprintf(stderr, "emu: exchange payload too large\n"); abort(); }
    Block* b = g_blk;
    const int lin = b->fibers[b->cur].linear;
    memcpy(&b->wv_buf[(size_t)lin * kSlotBytes], src, bytes);
    wave_sync();
}
const unsigned char* emu_wave_slot(int lane) {
    Block* b = g_blk;
    const int wave = b->fibers[b->cur].linear >> 6;
    return &b->wv_buf[((size_t)wave * 64 + lane) * kSlotBytes];
}
void emu_wave_exchange_end() { wave_sync(); }

void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    const int nwaves = (nthreads + 63) / 64;
    if (g_stacks_cap < (size_t)nthreads * kStack) {
        if (g_stacks) munmap(g_stacks, g_stacks_cap);
        g_stacks_cap = (size_t)nthreads * kStack;
        g_stacks = (char*)mmap(nullptr, g_stacks_cap, PROT_READ | PROT_WRITE,
                               MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (char*)MAP_FAILED) { perror("mmap"); abort(); }
    }
    gridDim = grid;
    blockDim = block;
    g_body = &body;
    Block blk;
    blk.nthreads = nthreads;
    blk.fibers.resize(nthreads);
    blk.wv_arrived.assign(nwaves, 0);
    blk.wv_gen.assign(nwaves, 0);
    blk.wv_buf.assign((size_t)nwaves * 64 * kSlotBytes, 0);
    g_blk = &blk;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = emu_uint3{bx, by, bz};
                blk.bar_arrived = 0;
                for (int w = 0; w < nwaves; ++w) blk.wv_arrived[w] = 0;
                int lin = 0;
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx, ++lin) {
                            Fiber& f = blk.fibers[lin];
                            f.tid = emu_uint3{tx, ty, tz};
                            f.linear = lin;
                            f.done = false;
                            getcontext(&f.ctx);
                            f.ctx.uc_stack.ss_sp = g_stacks + (size_t)lin * kStack;
                            f.ctx.uc_stack.ss_size = kStack;
                            f.ctx.uc_link = nullptr;
                            makecontext(&f.ctx, fiber_entry, 0);
                        }
                int remaining = nthreads;
                while (remaining > 0) {
                    for (int i = 0; i < nthreads; ++i) {
                        Fiber& f = blk.fibers[i];
                        if (f.done) continue;
                        blk.cur = i;
                        threadIdx = f.tid;
                        swapcontext(&g_sched, &f.ctx);
                        if (f.done) --remaining;
                    }
                }
            }
    g_blk = nullptr;
    g_body = nullptr;
}
