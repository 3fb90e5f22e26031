// hipemu runtime: fibers, the cooperative scheduler and the cross-lane operations (see hip/hip_runtime.h).
// Test infrastructure only.
#include "hip/hip_runtime.h"
#include <vector>
#include <dlfcn.h>

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
// x86-64 SysV: callee-saved registers on the outgoing stack, swap stack pointers, restore from the incoming stack
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
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
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {
namespace {

const size_t kStack = 256 << 10;

struct Wave {
    uint64_t slot[64];
    int arrived = 0, live = 0;
    unsigned gen = 0, res_gen = ~0u;
    uint64_t result = 0;
    int kind = 0;
    uint64_t seq = 0;
    uint64_t live_mask = 0;
};
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    dim3 tid;
    int lane = 0, wave = 0;
    uint64_t seq = 0;  // cross-lane operations executed by this lane
};
struct Block {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int live = 0, b_arrived = 0;
    int or_acc = 0;  // __syncthreads_or
    unsigned b_gen = 0;
    dim3 bid, bdim, gdim;
    const std::function<void()>* body = nullptr;
    Fiber* cur = nullptr;
    void* sched_sp = nullptr;
};
Block* g_blk = nullptr;
uint64_t g_coll = 0;
bool g_check_site = true;
unsigned g_group = 64;  // lanes that rendezvous with each other: 64, or the sub-wave group size of a kernel whose groups diverge freely
std::vector<char*> g_stacks;  // reused across launches

void yield() { hipemu_switch(&g_blk->cur->sp, g_blk->sched_sp); }

// A rendezvous is identified by the kind of operation and by how many cross-lane operations the lane has executed so
// far (return addresses do not work: the compiler duplicates calls along threaded jumps).  Lanes that pair up with
// different kinds or different counts took different paths through code that contains cross-lane operations.
void die(const char* what, const void* ra, long a, long b, long c, long d) {
    Dl_info ia;
    memset(&ia, 0, sizeof(ia));
    dladdr(ra, &ia);
    // the offset is relative to the shared object: `addr2line -e <so> <offset>` names the source line
    fprintf(stderr, "hipemu: %s (lane %d of wave %d, block %u; kind %ld/%ld, sequence %ld/%ld; arriving from %s+0x%lx)\n", what, g_blk->cur->lane,
            g_blk->cur->wave, g_blk->bid.x, a, b, c, d, ia.dli_fname ? ia.dli_fname : "?", (unsigned long)((const char*)ra - (const char*)ia.dli_fbase));
    abort();
}

// the caller waits until every live lane of its wave has arrived at the same call site
void wsync(const void* ra, int kind) {
    Block& B = *g_blk;
    Wave& w = B.waves[B.cur->wave];
    const uint64_t seq = ++B.cur->seq;
    if (w.arrived == 0) { w.kind = kind; w.seq = seq; }
    else if (g_check_site && (w.kind != kind || w.seq != seq)) die("divergent cross-lane operation", ra, w.kind, kind, (long)w.seq, (long)seq);
    const unsigned g = w.gen;
    if (++w.arrived >= w.live) {
        w.arrived = 0;
        w.gen++;
    } else {
        uint64_t spins = 0;
        while (w.gen == g) {
            yield();
            if (++spins == 100000) {  // every other lane had a hundred thousand turns and the rendezvous is still incomplete
                fprintf(stderr, "hipemu: deadlock: lane %d waits at kind %d, sequence %lu; %d of %d live lanes arrived\n", B.cur->lane, kind,
                        (unsigned long)seq, w.arrived, w.live);
                for (const Fiber& f : B.fibers) fprintf(stderr, " %d:%lu%s", f.lane, (unsigned long)f.seq, f.done ? "x" : "");
                fprintf(stderr, "\n");
                abort();
            }
        }
    }
}

void fiber_exit() {
    Block& B = *g_blk;
    Fiber* f = B.cur;
    Wave& w = B.waves[f->wave];
    f->done = true;
    w.slot[f->lane] = 0;
    w.live--;
    w.live_mask &= ~(1ull << f->lane);
    B.live--;
    if (w.arrived > 0 && w.arrived >= w.live) { w.arrived = 0; w.gen++; }
    if (B.b_arrived > 0 && B.b_arrived >= B.live) { B.b_arrived = 0; B.b_gen++; }
    yield();
    abort();  // a finished fiber is never resumed
}

void fiber_main() {
    (*g_blk->body)();
    fiber_exit();
}

}  // namespace

const dim3& thread_idx() { return g_blk->cur->tid; }
const dim3& block_idx() { return g_blk->bid; }
const dim3& block_dim() { return g_blk->bdim; }
const dim3& grid_dim() { return g_blk->gdim; }
int lane() { return g_blk->cur->lane; }
uint64_t collectives() { return g_coll; }
// Kernels that give each G-lane group of a wave its own work item (kc_zstd_match.hip: 8 units per wave) let the groups diverge; the
// hardware then runs them one after the other under the execution mask, and a ballot / shuffle of one group sees only its own
// lanes.  With set_group(G) the lanes of a group rendezvous among themselves (G must divide 64; 64 restores whole waves).
void set_group(unsigned g) { g_group = (g == 0 || g > 64 || (64 % g) != 0) ? 64 : g; }
// The order in which the scheduler's round robin visits the lanes: when several lanes of one instruction store to one address the
// hardware keeps ANY of them; forward order keeps the highest lane's value, reverse order the lowest lane's (tests of kernels
// that resolve such collisions themselves run under both).
static bool g_reverse = false;
void set_reverse(bool r) { g_reverse = r; }

// A wave that waits for another wave of its workgroup (a flag or a queue index in LDS, polled around s_sleep): no rendezvous, the
// lane just gives way; the scheduler's round robin runs every other fiber of the block before it returns here.  The caller's loop
// re-reads the flag after the call (the call is opaque to the compiler, like the hardware's volatile / atomic load).
__attribute__((noinline)) void pause() { yield(); }

__attribute__((noinline)) void wave_sync() {
    g_coll++;
    wsync(__builtin_return_address(0), 1);
}

__attribute__((noinline)) uint64_t ballot(bool p) {
    const void* site = __builtin_return_address(0);
    Block& B = *g_blk;
    Wave& w = B.waves[B.cur->wave];
    w.slot[B.cur->lane] = p ? 1 : 0;
    wsync(site, 2);
    if (w.res_gen != w.gen) {
        uint64_t m = 0;
        for (int i = 0; i < 64; i++) if ((w.live_mask >> i) & 1) m |= (w.slot[i] & 1) << i;
        w.result = m;
        w.res_gen = w.gen;
        g_coll++;
    }
    const uint64_t r = w.result;
    wsync(site, 3);
    return r;
}

__attribute__((noinline)) uint64_t shfl64(uint64_t v, int src_lane) {
    const void* site = __builtin_return_address(0);
    Block& B = *g_blk;
    Wave& w = B.waves[B.cur->wave];
    w.slot[B.cur->lane] = v;
    wsync(site, 4);
    const uint64_t r = w.slot[src_lane & 63];
    wsync(site, 5);
    if (B.cur->lane == 0) g_coll++;
    return r;
}

__attribute__((noinline)) uint64_t first_lane64(uint64_t v) {
    const void* site = __builtin_return_address(0);
    Block& B = *g_blk;
    Wave& w = B.waves[B.cur->wave];
    w.slot[B.cur->lane] = v;
    wsync(site, 6);
    const uint64_t r = w.slot[__builtin_ctzll(w.live_mask)];
    wsync(site, 7);
    return r;
}

__attribute__((noinline)) void block_sync() {
    const void* site = __builtin_return_address(0);
    Block& B = *g_blk;
    (void)site;
    const unsigned g = B.b_gen;
    if (++B.b_arrived >= B.live) {
        B.b_arrived = 0;
        B.b_gen++;
    } else {
        while (B.b_gen == g) yield();
    }
}

// __syncthreads_or: barrier + block-wide OR of the predicate
int block_or(int p) {
    block_sync();  // every fiber is past its reset of the previous use
    if (p) g_blk->or_acc = 1;
    block_sync();
    const int r = g_blk->or_acc;
    block_sync();  // every fiber has read
    g_blk->or_acc = 0;
    return r;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    if (const char* e = getenv("HIPEMU_NO_SITE_CHECK")) g_check_site = atoi(e) == 0;
    const unsigned nt = block.x * block.y * block.z;
    if (nt == 0 || nt > 1024) { fprintf(stderr, "hipemu: bad block size %u\n", nt); abort(); }
    while (g_stacks.size() < nt) {
        void* p = nullptr;
        if (posix_memalign(&p, 64, kStack) != 0) abort();
        g_stacks.push_back((char*)p);
    }
    Block B;
    B.bdim = block;
    B.gdim = grid;
    B.body = &body;
    Block* const saved = g_blk;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                B.bid = dim3(bx, by, bz);
                B.fibers.assign(nt, Fiber());
                B.waves.assign((nt + g_group - 1) / g_group, Wave());
                B.live = (int)nt;
                B.b_arrived = 0;
                B.or_acc = 0;
                for (unsigned t = 0; t < nt; t++) {
                    Fiber& f = B.fibers[t];
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.lane = (int)(t & 63);
                    f.wave = (int)(t / g_group);  // (lane numbers stay those of the 64-wide wave: ballot bits and shuffle sources are absolute)
                    f.stack = g_stacks[t];
                    Wave& w = B.waves[f.wave];
                    w.live++;
                    w.live_mask |= 1ull << f.lane;
                    // initial frame: six callee-saved registers, the entry point as return address, one pad slot
                    // (the entry sees rsp = 16n + 8, as after a call)
                    uint64_t* top = (uint64_t*)(((uintptr_t)f.stack + kStack) & ~(uintptr_t)15);
                    uint64_t* sp = top - 8;
                    for (int k = 0; k < 6; k++) sp[k] = 0;
                    sp[6] = (uint64_t)(uintptr_t)&fiber_main;
                    sp[7] = 0;
                    f.sp = sp;
                }
                g_blk = &B;
                while (B.live > 0) {
                    for (unsigned t0 = 0; t0 < nt; t0++) {
                        const unsigned t = g_reverse ? nt - 1 - t0 : t0;
                        Fiber& f = B.fibers[t];
                        if (f.done) continue;
                        B.cur = &f;
                        hipemu_switch(&B.sched_sp, f.sp);
                    }
                }
            }
    g_blk = saved;
}

}  // namespace hipemu
