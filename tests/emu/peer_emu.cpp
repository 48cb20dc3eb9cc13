// peer_emu.cpp -- csrc/peer_epoch.hpp between CPU processes (tests only; never linked into the product).
// The per-tick publish step of ndp_hip.hip's peer_publish_kernel + peer_epoch_kernel, run by one thread per process over POSIX shared memory:
// the SAME protocol text (PeerProto) on a CPU memory backend, so that world-size-2 gloo tests can check writer -> reader
// ordering, slot reuse and the bounded waits without a GPU.
#include <string.h>
#include <time.h>

#include "../../ndp_nmpc_qd_amd/csrc/peer_epoch.hpp"

using namespace ndp;

struct PeerCpuMem {
    typedef unsigned long long u64;
    static u64 load(const u64 *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
    static u64 peek(const u64 *p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
    static void store(u64 *p, u64 v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
    static u64 now_us()
    {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (u64)ts.tv_sec * 1000000ull + (u64)ts.tv_nsec / 1000ull;
    }
};

extern "C" {

size_t peer_emu_buffer_bytes(size_t n) { return peer_buffer_bytes(n); }
size_t peer_emu_slot_offset(size_t n, int s) { return peer_slot_offset(n, s); }

// one publish step (peer_publish_kernel with one "block", then peer_epoch_kernel); returns the tick it published
unsigned long long peer_emu_publish(const double *src, size_t n, void *own_buf, void *nb_buf, int slot, unsigned timeout_us)
{
    typedef PeerProto<PeerCpuMem> PP;
    typedef unsigned long long u64;
    u64 *own = (u64 *)own_buf, *nb = (u64 *)nb_buf;
    const u64 t = PP::next_tick(own);
    PP::ack_previous(nb, t);
    if (!PP::wait_slot_free(own, t, timeout_us)) own[PEER_W_STAT + PEER_STAT_ACK_TIMEOUT] += 1;
    memcpy((unsigned char *)own_buf + peer_slot_offset(n, (int)(t & 1)), src, n * 8);
    PP::set_epoch(own, t);          // release: the memcpy above is ordered before it
    own[PEER_W_STAT + PEER_STAT_TICKS] = t;
    if ((int)(t & 1) != slot) own[PEER_W_STAT + PEER_STAT_DESYNC] += 1;
    if (!PP::wait_epoch(nb, t, timeout_us)) own[PEER_W_STAT + PEER_STAT_EPOCH_TIMEOUT] += 1;
    return t;
}

void peer_emu_stats(const void *own_buf, unsigned long long *out4)
{
    for (int i = 0; i < PEER_STAT_N; ++i) out4[i] = ((const unsigned long long *)own_buf)[PEER_W_STAT + i];
}

}  // extern "C"
