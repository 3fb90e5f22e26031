// Self-test of tools/hipemu (the wave emulator the LDS-kernel tests stand on): cross-lane operations give the hardware's answers,
// lanes run out of lockstep between two of them, divergent collectives are caught.  argv[1]: "ok" | "diverge".
#include <hip/hip_runtime.h>
#include <string.h>

__global__ void k_ops(uint32_t* out) {
    __shared__ uint32_t sh[64];
    const int lane = (int)threadIdx.x;
    const uint64_t m = __ballot(lane % 3 == 0);
    const int up = __shfl_up(lane, 1, 8);      // width 8: lane 8k keeps its own value
    const int x = __shfl_xor(lane, 5, 64);
    const int rl = __builtin_amdgcn_readlane(lane * 7, 9);
    const int fl = __builtin_amdgcn_readfirstlane(lane + 100);
    sh[lane] = (uint32_t)lane * 2;
    hipemu::wave_sync();                        // KC_WAVE_SYNC in the kernels
    const uint32_t nb = sh[63 - lane];
    out[lane * 8 + 0] = (uint32_t)m;
    out[lane * 8 + 1] = (uint32_t)(m >> 32);
    out[lane * 8 + 2] = (uint32_t)up;
    out[lane * 8 + 3] = (uint32_t)x;
    out[lane * 8 + 4] = (uint32_t)rl;
    out[lane * 8 + 5] = (uint32_t)fl;
    out[lane * 8 + 6] = nb;
}

// no synchronisation between the store and the neighbour's load: on the emulator lane 0 runs to the end before lane 1 starts, so
// lane 0 reads what lane 1 has NOT yet written — the property that makes an unfenced LDS exchange fail on the CPU
__global__ void k_unfenced(uint32_t* out) {
    __shared__ uint32_t sh[64];
    const int lane = (int)threadIdx.x;
    sh[lane] = 0;
    hipemu::wave_sync();
    sh[lane] = 1000u + (uint32_t)lane;
    out[lane] = sh[(lane + 1) & 63];
}

__global__ void k_diverge(uint32_t* out) {
    const int lane = (int)threadIdx.x;
    uint64_t m = 0;
    if (lane < 32) m = __ballot(true);          // half the wave goes through a collective the other half skips ...
    const int v = __shfl(lane, 3);              // ... and meets it again at a different one
    out[lane] = (uint32_t)m + (uint32_t)v;
}

// Two waves of one workgroup, a one-way queue in LDS between them (the producer / consumer arrangement of a parse wave feeding an
// emit wave): wave 0 appends records and publishes the count, wave 1 polls the count around s_sleep and folds the records.  Needs
// hipemu::pause() — a wait that is neither a wave-wide nor a block-wide rendezvous.
#define QN 1000
#define QRING 64
__global__ void k_queue(uint32_t* out) {
    __shared__ uint32_t ring[QRING];
    __shared__ uint32_t head, tail;  // records published / records consumed
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    if (threadIdx.x == 0) { head = 0; tail = 0; }
    __syncthreads();
    if (wave == 0) {
        for (uint32_t base = 0; base < QN; base += 16) {  // 16 records per step, one per lane
            while (base + 16 > *(volatile uint32_t*)&tail + QRING) __builtin_amdgcn_s_sleep(1);  // ring full: wait for the consumer
            if (lane < 16 && base + lane < QN) ring[(base + lane) % QRING] = (base + lane) * 2654435761u;
            hipemu::wave_sync();  // the records before the count
            if (lane == 0) head = base + 16 < QN ? base + 16 : QN;
            hipemu::wave_sync();
        }
    } else {
        uint32_t acc = 0, done = 0;
        while (done < QN) {
            while (*(volatile uint32_t*)&head == done) __builtin_amdgcn_s_sleep(1);
            // one reading for the whole wave (the lanes poll out of lockstep here and could see different counts)
            const uint32_t h = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile uint32_t*)&head);
            for (uint32_t i = done + lane; i < h; i += 64) acc += ring[i % QRING];
            hipemu::wave_sync();  // every lane has read its records before the slots are handed back
            if (lane == 0) tail = h;
            done = h;
            hipemu::wave_sync();
        }
        out[lane] = acc;
    }
}

int main(int argc, char** argv) {
    static uint32_t out[64 * 8];
    uint32_t* o = out;
    if (argc > 1 && strcmp(argv[1], "diverge") == 0) {
        hipLaunchKernelGGL(k_diverge, dim3(1), dim3(64), 0, 0, o);
        printf("not reached\n");
        return 0;
    }
    hipLaunchKernelGGL(k_ops, dim3(1), dim3(64), 0, 0, o);
    uint64_t want = 0;
    for (int i = 0; i < 64; i += 3) want |= 1ull << i;
    for (int l = 0; l < 64; l++) {
        const uint32_t* r = out + l * 8;
        const int up = (l % 8) ? l - 1 : l;
        if (r[0] != (uint32_t)want || r[1] != (uint32_t)(want >> 32) || r[2] != (uint32_t)up || r[3] != (uint32_t)(l ^ 5) || r[4] != 63u || r[5] != 100u || r[6] != (uint32_t)(63 - l) * 2) {
            printf("lane %d: %u %u %u %u %u %u %u\n", l, r[0], r[1], r[2], r[3], r[4], r[5], r[6]);
            return 1;
        }
    }
    hipLaunchKernelGGL(k_unfenced, dim3(1), dim3(64), 0, 0, o);
    int stale = 0;
    for (int l = 0; l < 64; l++) stale += out[l] == 0u;   // lane l read its neighbour's slot before the neighbour ran
    // (all but one: the lane that completes the rendezvous runs on first, so one neighbour pair sees the new value)
    if (stale < 60) { printf("lanes did not run out of lockstep: %d stale\n", stale); return 2; }
    hipLaunchKernelGGL(k_queue, dim3(1), dim3(128), 0, 0, o);
    uint32_t got = 0, wantq = 0;
    for (int l = 0; l < 64; l++) got += out[l];
    for (uint32_t i = 0; i < QN; i++) wantq += i * 2654435761u;
    if (got != wantq) { printf("two-wave queue: %u vs %u\n", got, wantq); return 3; }
    printf("hipemu selftest ok\n");
    return 0;
}
