// grid_meet.h -- the meeting point of the blocks of a persistent launch (device code shared by
// the fp32 and the 16-bit kernel files; each includes it inside its own unnamed namespace).
#pragma once

// Meeting point of all blocks of a persistent launch.  It carries no data (nothing one block
// writes is read by another), so it needs no fences: its only purpose is TIMING -- the blocks'
// result bursts should hit HBM together, with nobody reading meanwhile (see the kernel below).
// The wait is bounded: a block that never shows up (not resident because something else holds
// its CU) costs the others a time-out once, after which they stop waiting; results never
// depend on it.
constexpr unsigned kSyncPolls = 96;

// Layout of a meeting counter: arrivals in the low kEpochShift bits, above them the tag (epoch)
// of the launch that counts in it.  Nothing is cleared between launches, neither by the host (a
// memset per launch costs more than a round of a small model's kernel) nor by the last block to
// leave: every block starts by looking at the tag, and the first one that finds another launch's
// tag -- or what an aborted or overlapping launch left behind -- restarts the word at (own tag,
// zero arrivals).  A counter therefore heals with the next launch that uses it, whatever it holds.
constexpr unsigned kEpochShift = 20;
constexpr unsigned kArriveMask = (1u << kEpochShift) - 1u;
constexpr unsigned kEpochs = (1u << (32 - kEpochShift)) - 1u;   // tags 1..kEpochs; 0 = never used

struct GridMeet {
    unsigned* word;
    unsigned arrivals;   // meetings this block has attended
    bool gave_up;
    unsigned blocks;     // blocks that meet on this word
};

// First access of every block of a phased launch to its counter.
__device__ __forceinline__ void grid_meet_enter(GridMeet& m, unsigned epoch) {
    if (threadIdx.x == 0) {
        unsigned cur = __hip_atomic_load(m.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // bounded: the swap only fails when another block changed the word meanwhile -- one of this
        // launch (then the tag matches) or of a launch sharing the slot (timing only; try a few times)
        for (int tries = 0; tries < 8 && (cur >> kEpochShift) != epoch; ++tries) {
            if (__hip_atomic_compare_exchange_strong(m.word, &cur, epoch << kEpochShift, __ATOMIC_RELAXED,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                break;
        }
    }
}

__device__ __forceinline__ void grid_meet(GridMeet& m) {
    ++m.arrivals;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(m.word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!m.gave_up) {
            // A meeting is over once HALF of the blocks of this round have arrived: waiting for
            // the stragglers costs more than their last reads beside the others' first writes
            // (measured, kernel ms with 0 / 32 / 128 / 192 of 256 blocks not waited for: 650 MB
            // vector 3.68 / 3.63 / 3.62 / 3.67, ResNet-50 K = 64 FedAvg 1.096 / 1.079 / 1.059 / 1.065).
            const unsigned target = m.arrivals * m.blocks - m.blocks / 2;
            unsigned polls = 0;
            while ((__hip_atomic_load(m.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kArriveMask) < target) {
                if (++polls >= kSyncPolls) {
                    m.gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
    }
    __syncthreads();
}

// Every block of a phased launch calls this exactly once, as its last access to the counter: a
// block that leaves before the others still counts for the meetings it skips.
__device__ __forceinline__ void grid_meet_leave(GridMeet& m, unsigned expected) {
    if (threadIdx.x == 0 && expected > m.arrivals)
        __hip_atomic_fetch_add(m.word, expected - m.arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
