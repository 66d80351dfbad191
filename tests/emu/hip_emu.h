// hip_emu.h — a minimal lane-level emulator for wave-synchronous device code (TEST HARNESS, x86-64 only).
//
// It runs one workgroup of 1..4 64-lane "wavefronts" as cooperatively scheduled fibers on one host thread (run_wave = one
// wavefront, run_block = several: barrier() is the workgroup barrier, ballot / shfl / gather64 / mfma are wave collectives).  A lane runs until it
// reaches a collective (barrier / ballot / shuffle), then the next lane runs; when all live lanes have arrived the
// collective completes.  That is enough to execute jvector_amd/csrc/gs_body.h — the body of the device-resident graph
// search kernel — unchanged on the CPU and compare it with the oracle, lane divergence, in-place compaction and
// hash-table races (as far as a deterministic schedule shows them) included.  What it cannot show: hardware memory
// ordering, register pressure, anything the real compiler does.  Not part of the product; nothing under jvector_amd/
// includes it.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace emu {

constexpr int WAVE = 64;
constexpr int MAX_WAVES = 4;
constexpr int MAXL = WAVE * MAX_WAVES;
constexpr size_t STACK_BYTES = 256 * 1024;

extern "C" void emu_switch(void **save_sp, void *next_sp);
// callee-saved registers + stack pointer swap (System V x86-64)
__asm__(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

// Lane scheduling order between collectives: lanes run one after another in the order order[0], order[1], ...
// EMU_LANE_ORDER = "reverse" or "random[:seed]" changes it (default: ascending).  A kernel that is correct under every
// order does not depend on which lane happens to run first inside a phase, i.e. every cross-lane hand-over goes through a
// collective or barrier — the property the real wavefront needs.
struct Wave {  // (a workgroup: nl = 64 * waves lanes; the name predates multi-wave blocks)
    int order[MAXL], pos_of[MAXL];
    void *sp[MAXL];
    char *stack[MAXL];
    bool done[MAXL];
    void *main_sp;
    int nl;
    int cur, live, arrived;
    unsigned gen;
    int wlive[MAX_WAVES], warrived[MAX_WAVES];  // wave-scope collectives
    unsigned wgen[MAX_WAVES];
    long long xbuf[MAXL];
    void (*fn)(void *);
    void *arg;
    long collectives;
};

inline Wave *&current()
{
    static Wave *w = nullptr;
    return w;
}
inline int lane() { return current()->cur; }            // index inside the workgroup (threadIdx.x)
inline int wave_id() { return current()->cur >> 6; }
inline int wave_lane() { return current()->cur & 63; }

inline int next_live_after(const Wave &w, int me)
{
    for (int i = 1; i <= w.nl; ++i) {
        const int c = w.order[(w.pos_of[me] + i) % w.nl];
        if (!w.done[c]) return c;
    }
    return me;
}

inline void switch_to_next_live()
{
    Wave &w = *current();
    const int me = w.cur;
    const int nx = next_live_after(w, me);
    if (nx == me) return;
    w.cur = nx;
    emu_switch(&w.sp[me], w.sp[nx]);
}

inline void barrier()
{
    Wave &w = *current();
    const unsigned g = w.gen;
    if (++w.arrived >= w.live) {
        w.arrived = 0;
        w.gen++;
        w.collectives++;
        return;
    }
    while (w.gen == g) switch_to_next_live();
}

// barrier among the live lanes of the calling lane's wavefront (the scope of ballot / shuffle / mfma)
inline void wave_barrier()
{
    Wave &w = *current();
    const int wv = w.cur >> 6;
    const unsigned g = w.wgen[wv];
    if (++w.warrived[wv] >= w.wlive[wv]) {
        w.warrived[wv] = 0;
        w.wgen[wv]++;
        w.collectives++;
        return;
    }
    while (w.wgen[wv] == g) switch_to_next_live();
}

inline void lane_exit()
{
    Wave &w = *current();
    const int me = w.cur;
    w.done[me] = true;
    w.xbuf[me] = 0;
    w.live--;
    w.wlive[me >> 6]--;
    if (w.live == 0) {
        void *dummy;
        emu_switch(&dummy, w.main_sp);  // never returns
    }
    if (w.arrived >= w.live) {  // the others were waiting for this lane only
        w.arrived = 0;
        w.gen++;
    }
    if (w.wlive[me >> 6] > 0 && w.warrived[me >> 6] >= w.wlive[me >> 6]) {
        w.warrived[me >> 6] = 0;
        w.wgen[me >> 6]++;
    }
    const int nx = next_live_after(w, me);
    w.cur = nx;
    void *dummy;
    emu_switch(&dummy, w.sp[nx]);  // never returns
}

inline void lane_entry()
{
    Wave &w = *current();
    w.fn(w.arg);
    lane_exit();
    abort();
}

// run fn(arg) on a workgroup of `waves` wavefronts until every lane has returned
inline long run_block(void (*fn)(void *), void *arg, int waves)
{
    Wave *w = new Wave();
    memset(w, 0, sizeof(*w));
    w->fn = fn;
    w->arg = arg;
    const int NL = WAVE * waves;
    w->nl = NL;
    w->live = NL;
    for (int v = 0; v < waves; ++v) w->wlive[v] = WAVE;
    for (int i = 0; i < NL; ++i) {
        w->stack[i] = (char *)aligned_alloc(64, STACK_BYTES);
        uintptr_t top = ((uintptr_t)(w->stack[i] + STACK_BYTES)) & ~(uintptr_t)15;
        void **sp = (void **)top;
        *--sp = nullptr;                 // fake return address of lane_entry (keeps rsp % 16 == 8 at entry)
        *--sp = (void *)&lane_entry;     // `ret` target of the first switch
        for (int r = 0; r < 6; ++r) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
        w->sp[i] = (void *)sp;
    }
    {
        for (int i = 0; i < NL; ++i) w->order[i] = i;
        const char *e = getenv("EMU_LANE_ORDER");
        if (e && !strncmp(e, "reverse", 7)) {
            for (int i = 0; i < NL; ++i) w->order[i] = NL - 1 - i;
        } else if (e && !strncmp(e, "random", 6)) {
            static unsigned long long state = 0;
            if (!state) state = e[6] == ':' ? strtoull(e + 7, nullptr, 10) * 2654435761ull + 1 : 88172645463325252ull;
            for (int i = NL - 1; i > 0; --i) {  // Fisher-Yates with xorshift64, a fresh permutation per wave
                state ^= state << 13;
                state ^= state >> 7;
                state ^= state << 17;
                const int j = (int)(state % (unsigned long long)(i + 1));
                const int t = w->order[i];
                w->order[i] = w->order[j];
                w->order[j] = t;
            }
        }
        for (int i = 0; i < NL; ++i) w->pos_of[w->order[i]] = i;
    }
    Wave *prev = current();
    current() = w;
    w->cur = w->order[0];
    emu_switch(&w->main_sp, w->sp[w->order[0]]);
    current() = prev;
    const long n = w->collectives;
    for (int i = 0; i < NL; ++i) free(w->stack[i]);
    delete w;
    return n;
}
inline long run_wave(void (*fn)(void *), void *arg) { return run_block(fn, arg, 1); }

inline uint64_t ballot(bool p)
{
    Wave &w = *current();
    const int base = w.cur & ~63;
    w.xbuf[w.cur] = p ? 1 : 0;
    wave_barrier();
    uint64_t m = 0;
    for (int i = 0; i < WAVE; ++i)
        if (!w.done[base + i] && w.xbuf[base + i]) m |= 1ull << i;
    wave_barrier();
    return m;
}
inline long long shfl(long long v, int src)
{
    Wave &w = *current();
    const int base = w.cur & ~63;
    w.xbuf[w.cur] = v;
    wave_barrier();
    const long long r = w.xbuf[base + (src & 63)];
    wave_barrier();
    return r;
}


// every lane receives the 64 values of the wave (lanes that have exited contribute 0)
inline void gather64(float v, float (&out)[WAVE])
{
    Wave &w = *current();
    static float G[MAXL];  // one workgroup runs at a time
    const int base = w.cur & ~63;
    G[w.cur] = v;
    wave_barrier();
    for (int i = 0; i < WAVE; ++i) out[i] = w.done[base + i] ? 0.0f : G[base + i];
    wave_barrier();
}

// v_mfma_f32_32x32x2_f32 as the hardware defines it (cdna_hip_programming.md §3): lane l supplies A[i = l & 31][k = l >> 5]
// and B[k = l >> 5][j = l & 31]; lane l's accumulator register r holds D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31];
// D = fma(A[i][1], B[1][j], fma(A[i][0], B[0][j], C)) — a k-ordered f32 fmaf chain, no wider internal accumulation.
struct f32x16 {
    float v[16];
    float &operator[](int i) { return v[i]; }
    const float &operator[](int i) const { return v[i]; }
};
inline f32x16 mfma_32x32x2(float a, float b, f32x16 c)
{
    Wave &w = *current();
    static float A[MAXL], B[MAXL];  // one workgroup runs at a time; each wavefront uses its own 64 slots
    const int base = w.cur & ~63;
    A[w.cur] = a;
    B[w.cur] = b;
    wave_barrier();
    const int l = w.cur & 63, j = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        c.v[r] = fmaf(A[base + i + 32], B[base + j + 32], fmaf(A[base + i], B[base + j], c.v[r]));
    }
    wave_barrier();
    return c;
}

}  // namespace emu
