// shim_runtime.cpp -- host execution of CUDA-style kernels for oracle/_ref (TEST INFRASTRUCTURE).
//
// One OS thread.  The blocks of a launch run one after another; the threads of a block are ucontext
// fibers.  A fiber that reaches a barrier switches back to the scheduler, which resumes it only
// after every still-running fiber of the block has arrived, so __syncthreads(),
// __syncthreads_count() and cooperative_groups::thread_block::sync() mean what they mean on a GPU.
// `static` variables standing in for __shared__ memory are naturally per block because blocks do
// not overlap in time.
#include "cuda_shim.h"

#include <ucontext.h>

#include <vector>

dim3 gridDim, blockDim;
uint3 blockIdx, threadIdx;

namespace gsr_shim {
namespace {
constexpr size_t kStackBytes = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    uint3 tid;
    bool finished = false;
    int predicate = 0;
};

ucontext_t g_scheduler;
Fiber* g_current = nullptr;
const std::function<void()>* g_body = nullptr;
int g_barrier_count = 0;
std::vector<char> g_stacks;

void fiber_main() {
    (*g_body)();
    g_current->finished = true;  // returning resumes uc_link == the scheduler
}
} // namespace

int barrier(int predicate) {
    Fiber* self = g_current;
    self->predicate = predicate;
    swapcontext(&self->ctx, &g_scheduler);
    return g_barrier_count;  // set by the scheduler before anyone is resumed
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const size_t nthreads = (size_t)block.x * block.y * block.z;
    if (nthreads == 0) return;
    gridDim = grid;
    blockDim = block;
    g_body = &body;
    if (g_stacks.size() < nthreads * kStackBytes) g_stacks.resize(nthreads * kStackBytes);
    std::vector<Fiber> fibers(nthreads);

    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = uint3{bx, by, bz};
                size_t t = 0;
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx, ++t) {
                            Fiber& f = fibers[t];
                            f.tid = uint3{tx, ty, tz};
                            f.finished = false;
                            f.predicate = 0;
                            getcontext(&f.ctx);
                            f.ctx.uc_stack.ss_sp = g_stacks.data() + t * kStackBytes;
                            f.ctx.uc_stack.ss_size = kStackBytes;
                            f.ctx.uc_link = &g_scheduler;
                            makecontext(&f.ctx, fiber_main, 0);
                        }
                size_t running = nthreads;
                while (running > 0) {
                    int count = 0;
                    size_t still = 0;
                    for (size_t i = 0; i < nthreads; ++i) {
                        Fiber& f = fibers[i];
                        if (f.finished) continue;
                        g_current = &f;
                        threadIdx = f.tid;
                        swapcontext(&g_scheduler, &f.ctx);  // runs until the next barrier or the end
                        if (!f.finished) {
                            ++still;
                            count += f.predicate != 0;
                        }
                    }
                    g_barrier_count = count;  // every unfinished fiber now waits at the same barrier
                    running = still;
                }
            }
    g_body = nullptr;
    g_current = nullptr;
}

} // namespace gsr_shim
