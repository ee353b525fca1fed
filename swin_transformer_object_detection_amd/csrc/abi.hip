#include <functional>
#include <vector>

#include "common.h"
#include <cstdlib>
extern "C" int swin_hip_abi_version(void) { return 2; }
extern "C" int swin_hip_half_type(void) {
#ifdef SWIN_HALF
    return 1;
#else
    return 0;
#endif
}

// ---- second stream for work that nothing on the main stream waits for ------------------------------------------------------
// Per device: while an auxiliary stream is set, the entry points that end in a small REDUCTION nobody on the main stream
// consumes -- swin_layernorm_bwd's parameter-gradient reduce, swin_window_attn_bwd's bias-gradient slab reduce,
// swin_rel_bias_reduce -- enqueue that launch on the auxiliary stream, behind an event recorded on the main stream after the
// producing kernel.  The caller joins the streams before the results are read, and must not hand the workspaces of those
// calls to later main-stream work before the join (the reductions still read them).
static void* g_aux_stream[16] = {};

extern "C" int swin_set_aux_stream(void* side) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    g_aux_stream[dev] = side;
    return SWIN_OK;
}

void* swin_aux_stream(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    return g_aux_stream[dev];
}

// Deferred form (used by swin_block_bwd): instead of one fork per reduction -- two runtime calls each, eight per Swin block --
// the reductions are collected while the block's data-gradient chain is enqueued and flushed behind ONE fork at its end.
static bool g_aux_defer[16] = {};
static std::vector<std::function<int(void*)>> g_aux_deferred[16];

static int cur_dev() {
    int dev = 0;
    return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) ? dev : -1;
}

void swin_aux_defer(bool on) {
    const int dev = cur_dev();
    if (dev < 0) return;
    g_aux_defer[dev] = on;
    if (!on) g_aux_deferred[dev].clear();
}

// true: `launch` was queued for swin_aux_flush; false: the caller launches it itself (forking if an auxiliary stream is set)
bool swin_aux_push(std::function<int(void*)> launch) {
    const int dev = cur_dev();
    if (dev < 0 || !g_aux_defer[dev] || !g_aux_stream[dev]) return false;
    g_aux_deferred[dev].push_back(std::move(launch));
    return true;
}

int swin_fork_stream(void* main, void* side);

// the queued launches, on `side` behind one fork from `main` (on `main` itself when side is null)
int swin_aux_flush(void* main, void* side) {
    const int dev = cur_dev();
    if (dev < 0) return SWIN_ERR_UNSUPPORTED;
    void* st = side ? side : main;
    if (side && side != main) {
        int f = swin_fork_stream(main, side);
        if (f != SWIN_OK) return f;
    }
    int rc = SWIN_OK;
    for (auto& fn : g_aux_deferred[dev]) {
        int r = fn(st);
        if (r != SWIN_OK && rc == SWIN_OK) rc = r;
    }
    g_aux_deferred[dev].clear();
    return rc;
}

// `side` waits for everything enqueued on `main` so far (no host synchronisation).  A ring of events per device:
// hipStreamWaitEvent captures the event's latest record at the time of the call, so an event may be re-recorded as soon as
// the wait has been enqueued; the ring only keeps that property from mattering.
extern "C" int swin_fork_stream(void* main, void* side) {
    // a NULL handle is the legacy default stream -- a valid stream on either side.  (An earlier `!side` early-out turned every
    // JOIN of the default stream into a no-op: side_join() passes the current stream as `side`, and torch's default stream IS
    // handle 0.  The step then ran with no join at all and still passed every single-step comparison by timing.)
    if (side == main) return SWIN_OK;
    // 2048 events per device: several steps of forks.  Re-recording an event whose previous record the GPU has not passed yet
    // makes the host wait for it on this runtime -- with a ring of 64 the host could never run more than half a step ahead.
    constexpr unsigned RING = 2048;
    static hipEvent_t ring[16][RING];
    static unsigned next_ev[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    hipEvent_t& ev = ring[dev][next_ev[dev]++ & (RING - 1)];
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return SWIN_ERR_LAUNCH;
    if (hipEventRecord(ev, (hipStream_t)main) != hipSuccess) return SWIN_ERR_LAUNCH;
    if (hipStreamWaitEvent((hipStream_t)side, ev, 0) != hipSuccess) return SWIN_ERR_LAUNCH;
    return SWIN_OK;
}

// A stream of the LOWEST priority the device offers (never destroyed: one or two per process).  The second stream carries work
// nothing waits for; at equal priority its workgroups take CUs and memory bandwidth from the data-gradient chain on the main
// stream, which is the step's critical path.
extern "C" int swin_stream_create_low_priority(void** out) {
    if (!out) return SWIN_ERR_BAD_ARG;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return SWIN_ERR_LAUNCH;
    hipStream_t s = nullptr;
    // experiment (SWIN_SIDE_CU_DROP=D): a stream that may not use every D-th CU instead of a low-priority one -- priorities do not
    // preempt, so a short kernel of the main stream otherwise waits for CUs behind 40-140 us weight-gradient workgroups
    static const int drop = swin_dev_int("SWIN_SIDE_CU_DROP", 0);
    if (drop >= 2) {
        uint32_t mask[16];
        for (int w = 0; w < 16; ++w) {
            uint32_t m = 0;
            for (int b = 0; b < 32; ++b) if ((w * 32 + b) % drop != drop - 1) m |= 1u << b;
            mask[w] = m;
        }
        if (hipExtStreamCreateWithCUMask(&s, 16, mask) != hipSuccess) return SWIN_ERR_LAUNCH;
        *out = (void*)s;
        return SWIN_OK;
    }
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) != hipSuccess) return SWIN_ERR_LAUNCH;
    *out = (void*)s;
    return SWIN_OK;
}
