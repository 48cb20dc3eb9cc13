// peer_epoch.hpp -- ordering protocol of the per-tick neighbour exchange through a peer-mapped window buffer.
//
// Replaces the ordering a ROS topic gives the reference for free: every control tick each vehicle publishes a NEW PredXU
// (its 21x10 float64 reference window, nmpc_node.py:116-133,229-230) and the leader's subscriber callback consumes the latest
// one (ndp_nmpc_leader_node.py:40,60-76).  One process per GPU: the publisher ("owner") keeps TWO window slots in a buffer
// whose IPC handle it gave to the subscriber ("reader") once; the reader's control-step kernel reads the slot straight out of
// the owner's HBM over xGMI.  What has to be ordered, per tick t = 1, 2, ... (slot s = t & 1):
//     owner :  wait until the reader is done with tick t-2 (the slot's previous content)      ack[s]   >= t - 2
//              write the windows of tick t into slot s, make them visible system-wide
//              epoch[s] := t                                                                    (release)
//     reader:  ack[(t-1) & 1] := t - 1   -- its control step of tick t-1 has finished, that slot may be overwritten
//              wait until epoch[s] >= t                                                         (acquire)
//              read slot s (its control-step kernel, launched next in stream order)
// Every rank is both (it publishes its own windows and reads the next rank's), in lockstep tick numbering; the tick number is
// not a kernel argument but read from the rank's own epoch words (t = max(epoch[0], epoch[1]) + 1), so a captured hipGraph of
// publish + control-step launches replays correctly.  Single reader per buffer (ring of ranks: rank r reads rank r+1).
//
// Waits are bounded (timeout_us): a reader whose publisher has gone away proceeds with the slot as it is -- the reference's
// subscriber likewise keeps using the last PredXU it received -- and an owner whose reader is stuck overwrites the slot, as a
// ROS publisher with a full queue drops the oldest message.  Both are counted (PEER_STAT_*), never silent.
//
// The text below is written against a small memory backend M (system-scope acquire loads / release stores, relaxed polls + a microsecond
// clock) so that tests/emu can run the very same protocol between two CPU processes over POSIX shared memory (gloo ranks).
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifndef NDP_PEER_FN
#define NDP_PEER_FN inline
#endif

namespace ndp {

// buffer layout (bytes): [0, 256) words shared with the reader | [256, 512) the owner's private words | slot 0 | slot 1
enum { PEER_HDR_BYTES = 512 };
enum { PEER_W_EPOCH = 0 /* + 8 s */, PEER_W_ACK = 16 /* + 8 s */, PEER_W_DONE = 32, PEER_W_STAT = 40 /* + i */ };
enum { PEER_STAT_TICKS = 0, PEER_STAT_ACK_TIMEOUT = 1, PEER_STAT_EPOCH_TIMEOUT = 2, PEER_STAT_DESYNC = 3, PEER_STAT_N = 4 };

NDP_PEER_FN size_t peer_slot_bytes(size_t n_doubles) { return (n_doubles * 8 + 255) & ~(size_t)255; }
NDP_PEER_FN size_t peer_buffer_bytes(size_t n_doubles) { return PEER_HDR_BYTES + 2 * peer_slot_bytes(n_doubles); }
NDP_PEER_FN size_t peer_slot_offset(size_t n_doubles, int s) { return PEER_HDR_BYTES + (size_t)s * peer_slot_bytes(n_doubles); }

template <class M>
struct PeerProto {
    typedef unsigned long long u64;

    // the tick this publish call is about: one more than the newest tick in the owner's own slots.  (M::peek: a load that orders
    // nothing behind it -- what follows these polls are the owner's own WRITES; an acquire load costs a cache invalidation per poll,
    // and every wave of the publish launch polls)
    static NDP_PEER_FN u64 next_tick(const u64 *own)
    {
        const u64 e0 = M::peek(own + PEER_W_EPOCH), e1 = M::peek(own + PEER_W_EPOCH + 8);
        return (e0 > e1 ? e0 : e1) + 1;
    }

    // reader role, start of tick t: the control step of tick t-1 is over (stream order), its slot may be overwritten
    static NDP_PEER_FN void ack_previous(u64 *nb, u64 t)
    {
        if (t > 1) M::store(nb + PEER_W_ACK + 8 * ((t - 1) & 1), t - 1);
    }

    // owner role: the reader must be done with tick t-2 before slot t & 1 is overwritten.  false = timed out (overwrites anyway).
    static NDP_PEER_FN bool wait_slot_free(const u64 *own, u64 t, unsigned timeout_us)
    {
        if (t <= 2) return true;
        const u64 *w = own + PEER_W_ACK + 8 * (t & 1);
        if (M::peek(w) + 2 >= t) return true;
        const u64 t0 = M::now_us();
        while (M::peek(w) + 2 < t)
            if (M::now_us() - t0 > timeout_us) return false;
        return true;
    }

    // owner role: after the slot's content is visible system-wide
    static NDP_PEER_FN void set_epoch(u64 *own, u64 t) { M::store(own + PEER_W_EPOCH + 8 * (t & 1), t); }

    // reader role: the neighbour's windows of tick t are in place.  false = timed out (the caller reads the slot as it is).
    static NDP_PEER_FN bool wait_epoch(const u64 *nb, u64 t, unsigned timeout_us)
    {
        const u64 *w = nb + PEER_W_EPOCH + 8 * (t & 1);
        if (M::load(w) >= t) return true;
        const u64 t0 = M::now_us();
        while (M::load(w) < t)
            if (M::now_us() - t0 > timeout_us) return false;
        return true;
    }
};

}  // namespace ndp
