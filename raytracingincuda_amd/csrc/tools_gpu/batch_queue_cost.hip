// batch_queue_cost.hip -- what does a barrier-free, LDS-resident batch queue cost per batch of 64 paths?   (VERDICT r04 "next" #2)
//
// The render kernel keeps a path in the registers of one lane; its divergent blocks run with ~26 of 64 lanes enabled
// (DESIGN.md 4.5).  The one regrouping design that needs no workgroup barrier keeps the paths RESIDENT IN LDS and lets the
// waves of a workgroup be interchangeable workers on phase-homogeneous batches: a wave pops 64 path records from the queue
// of one phase, runs that phase for all 64 with every lane enabled, writes the records back and pushes every path to the
// queue of its next phase.  DESIGN.md costed that on paper for two rounds.  This probe MEASURES the part the paper figure
// guessed -- the queue round trip -- on the real part, at the occupancy such a kernel can have:
//
//   * one workgroup per CU (its LDS is the path store): 1024 records of 96 B (24 words, AoS: six ds_read_b128 / ds_write_b128
//     per lane) + four phase queues (rings of 2048 u16 indices with a valid bit) + head / tail counters = 112 KB of LDS;
//   * 8 / 12 / 16 waves per workgroup = 2 / 3 / 4 waves per SIMD;
//   * a wave's trip: pick the fullest queue (two ds_read_b128 of the counters), claim <= 64 entries with ONE LDS
//     compare-and-swap on the queue's head (lane 0, retried on contention), read the indices (spin on the valid bit of an entry
//     whose pusher has reserved but not yet written it), clear them, load the 24-word records, run the STAND-IN BODY -- BODY
//     dependent v_fma_f32 in two interleaved chains on the loaded words --, store the records, choose each path's next phase
//     (a hash of its state: uniform over the four phases, so a wave's 64 paths scatter over all queues, as the phases of a
//     path tracer do), reserve ring slots with one LDS atomic add per phase present (ballot + mbcnt ranks), write the indices;
//   * a path retires after TRIPS batches; the workgroup ends when all 1024 have retired (a wave that finds every queue empty
//     sleeps and polls -- part of the price);
//   * the BASELINE is the same number of body executions with the state in registers and no queue: waves x (1024 x TRIPS / 64
//     / waves) bodies.  Queue overhead per batch = (t_queue - t_baseline) x clock / batches per SIMD, in SIMD cycles, and in
//     instruction-equivalents (/ 3.87, the measured issue cost of a plain v_fma_f32 stream at five waves per SIMD:
//     profiles/archive/r01_valu_cost.json).
//
// GO / NO-GO RULE, fixed before the first run (committed with this file, before any number existed):
//   GO   if at 3 waves per SIMD the queue overhead per batch is <= 25 % of the 100-instruction body's own cost
//        (i.e. <= 25 instruction-equivalents ~ 100 SIMD cycles per batch) AND the mean batch fill is >= 48 of 64 lanes;
//   NO-GO otherwise: with ~4.6 batch executions per path segment (hit_world, sky / hit record, the two rejection loops, the
//        material tails) a larger overhead eats what full lanes save -- today's kernel spends ~13 wave-instructions per
//        segment and lane, the batch form ~5.4 of work + 4.6 x overhead / 64 x ... see DESIGN.md section 8 for the budget.
// On GO the three rejection / material phases of the render kernel move to batches and are A/B-timed; on NO-GO DESIGN.md
// section 8 is rewritten from these numbers and the design is closed.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o batch_queue_cost batch_queue_cost.hip          Output: one JSON object.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int NREC = 1024, WORDS = 24, NQ = 4, TRIPS = 48;
constexpr int RING = 2 * NREC;          // twice the records: a slot cannot come round again while its last entry is claimed but not yet cleared
constexpr int GUARD = 200000;            // loop trips a wave may take at most (~150 expected): the kernel ends even if the queue logic were wrong
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

struct Lds {
    u4 rec[NREC][WORDS / 4];            // 96 B per path
    uint16_t ring[NQ][RING];            // entry = index | 0x8000 once written, 0 = empty
    uint32_t head[NQ], tail[NQ];        // monotonically increasing positions (mod NREC in the ring)
    uint32_t retired;
};

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <int BODY>
__device__ __forceinline__ void body(float (&w)[WORDS]) {
    // BODY dependent vector instructions: two chains of BODY / 2 (a path-tracing phase has about that much independence)
    float a = w[0], b = w[1];
    const float c = w[2], d = w[3];
#pragma unroll
    for (int k = 0; k < BODY / 2; ++k) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b) : "v"(d), "v"(c));
    }
    w[0] = a; w[1] = b;
}

// counters[0] batches, [1] records processed, [2] pop attempts that found every queue empty, [3] CAS retries, [4] valid-bit spins, [5] waves stopped by GUARD (must be 0)
template <int BODY, bool STAMPS>
__global__ void __launch_bounds__(1024) queue_kernel(unsigned long long* counters, float* sink) {
#define STAMP() (STAMPS ? __builtin_amdgcn_s_memtime() : 0ull)
#define STAMP_LANE(v) do { if (STAMPS) v = __builtin_amdgcn_s_memtime(); } while (0)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS-qualified view: every access below must be a ds_* instruction.  (Run 1 of this probe went through a generic `Lds&`: its volatile
    // reads of the counters and ring entries compiled to flat_load / flat_store on the LDS aperture, 4000+ cycles per batch --
    // profiles/r05/batch_queue_cost_run1_flat_ops.json, kept as a record of the mistake, not as a result.)
    typedef __attribute__((address_space(3))) Lds LdsT;
    LdsT& L = *(LdsT*)smem;
    const int lane = lane_id();
    for (int r = threadIdx.x; r < NREC; r += blockDim.x) {
        for (int k = 0; k < WORDS / 4; ++k) {
            const float f = 0.5f + 1e-4f * (float)((r * 7 + k * 13) & 1023);
            u4 v; v.x = __float_as_uint(f); v.y = __float_as_uint(f * 0.9f); v.z = __float_as_uint(0.999f); v.w = __float_as_uint(1e-3f);
            L.rec[r][k] = v;
        }
        L.rec[r][5].w = TRIPS;                                   // word 23: batches left
        L.rec[r][5].z = (uint32_t)r * 2654435761u;               // word 22: the path's "generator": drives its phase sequence
        L.ring[0][r] = (uint16_t)(r | 0x8000); L.ring[0][r + NREC] = 0;
        for (int q = 1; q < NQ; ++q) { L.ring[q][r] = 0; L.ring[q][r + NREC] = 0; }
    }
    if (threadIdx.x < NQ) { L.head[threadIdx.x] = 0; L.tail[threadIdx.x] = threadIdx.x == 0 ? NREC : 0; }
    if (threadIdx.x == 0) L.retired = 0;
    __syncthreads();                                             // the ONLY barrier: before the first batch
    unsigned long long n_batches = 0, n_records = 0, n_empty = 0, n_retry = 0, n_spin = 0;
    unsigned long long c_pick = 0, c_load = 0, c_body = 0, c_store = 0, c_push = 0;      // s_memtime cycles of a wave per section (STAMPS builds)
    float keep = 0;
    unsigned long long n_guard = 0;
    for (int trip = 0;; ++trip) {
        if (trip >= GUARD) { n_guard = 1; break; }
        const unsigned long long s0 = STAMP();
        // ---- pick the fullest queue and claim up to 64 of its entries: lane 0, one compare-and-swap
        int q = 0, n = 0;
        uint32_t h = 0;
        if (lane == 0) {
            bool got = false;
            for (int tries = 0; tries < 64 && !got; ++tries) {
                n = 0;
                const u4 hd = *(volatile __attribute__((address_space(3))) u4*)L.head, tl = *(volatile __attribute__((address_space(3))) u4*)L.tail;
                const uint32_t av[NQ] = {tl.x - hd.x, tl.y - hd.y, tl.z - hd.z, tl.w - hd.w};
                const uint32_t hs[NQ] = {hd.x, hd.y, hd.z, hd.w};
                q = 0;
                for (int k = 1; k < NQ; ++k) if (av[k] > av[q]) q = k;
                n = av[q] < 64u ? (int)av[q] : 64;
                if (n == 0) break;
                h = hs[q];
                uint32_t expect = h;
                if (__hip_atomic_compare_exchange_strong(&L.head[q], &expect, h + (uint32_t)n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) got = true;
                else ++n_retry;
            }
            if (!got) n = 0;                                      // every queue empty, or 64 lost races in a row: poll again
        }
        n = __builtin_amdgcn_readfirstlane(n); q = __builtin_amdgcn_readfirstlane(q); h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
        if (n == 0) {
            ++n_empty;
            if (__hip_atomic_load(&L.retired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (uint32_t)NREC) break;
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        ++n_batches; n_records += (unsigned)n;
        const unsigned long long s1 = STAMP();
        // ---- the batch: indices, records
        const bool on = lane < n;
        unsigned long long s2 = 0, s3 = 0;
        int idx = 0;
        float w[WORDS];
        if (on) {
            volatile __attribute__((address_space(3))) uint16_t* e = &L.ring[q][(h + (uint32_t)lane) & (RING - 1)];
            uint16_t v;
            while (((v = *e) & 0x8000) == 0 && n_spin < 1000000ull) ++n_spin;   // reserved by its pusher, not written yet (bounded: see GUARD)
            *e = 0;
            idx = v & 0x7fff;
#pragma unroll
            for (int k = 0; k < WORDS / 4; ++k) {
                const u4 v4 = L.rec[idx][k];
                w[4 * k] = __uint_as_float(v4.x); w[4 * k + 1] = __uint_as_float(v4.y); w[4 * k + 2] = __uint_as_float(v4.z); w[4 * k + 3] = __uint_as_float(v4.w);
            }
            STAMP_LANE(s2);
            body<BODY>(w);
            STAMP_LANE(s3);
        }
        // ---- next phase of every path, records back, push
        int nq = -1;
        if (on) {
            uint32_t g = __float_as_uint(w[22]), left = __float_as_uint(w[23]) - 1u;
            g = g * 1664525u + 1013904223u;
            w[22] = __uint_as_float(g); w[23] = __uint_as_float(left);
#pragma unroll
            for (int k = 0; k < WORDS / 4; ++k) {
                u4 v4; v4.x = __float_as_uint(w[4 * k]); v4.y = __float_as_uint(w[4 * k + 1]); v4.z = __float_as_uint(w[4 * k + 2]); v4.w = __float_as_uint(w[4 * k + 3]);
                L.rec[idx][k] = v4;
            }
            nq = left == 0 ? -1 : (int)(g >> 30);
            keep += w[0];
        }
        const unsigned long long s4 = STAMP();
        const unsigned long long m_ret = __builtin_amdgcn_ballot_w64(on && nq < 0);
        if (m_ret != 0 && lane == 0) __hip_atomic_fetch_add(&L.retired, (uint32_t)__builtin_popcountll(m_ret), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(nq == k);
            if (m == 0) continue;
            uint32_t base = 0;
            if (lane == 0) base = __hip_atomic_fetch_add(&L.tail[k], (uint32_t)__builtin_popcountll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (nq == k) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                L.ring[k][(base + rank) & (RING - 1)] = (uint16_t)(idx | 0x8000);
            }
        }
        if (STAMPS) {
            const unsigned long long s5 = STAMP();
            s2 = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)s2) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(s2 >> 32)) << 32);
            s3 = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)s3) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(s3 >> 32)) << 32);
            c_pick += s1 - s0; c_load += s2 - s1; c_body += s3 - s2; c_store += s4 - s3; c_push += s5 - s4;
        }
    }
    if (keep == 12345.678f) sink[0] = keep;
    if (lane == 0) {
        atomicAdd(&counters[0], n_batches); atomicAdd(&counters[1], n_records); atomicAdd(&counters[2], n_empty);
        atomicAdd(&counters[3], n_retry); atomicAdd(&counters[5], n_guard);
        if (STAMPS) { atomicAdd(&counters[6], c_pick); atomicAdd(&counters[7], c_load); atomicAdd(&counters[8], c_body); atomicAdd(&counters[9], c_store); atomicAdd(&counters[10], c_push); }
    }
    unsigned long long s = n_spin;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) atomicAdd(&counters[4], s);
}

// The same bodies with the state in registers and nothing else: `per_wave` full batches per wave.
template <int BODY>
__global__ void __launch_bounds__(1024) baseline_kernel(int per_wave, float* sink) {
    float w[WORDS];
    for (int k = 0; k < WORDS; ++k) w[k] = 0.5f + 1e-4f * (float)((threadIdx.x * 7 + k * 13) & 1023);
    w[2] = 0.999f; w[3] = 1e-3f;
    for (int b = 0; b < per_wave; ++b) {
        body<BODY>(w);
        asm volatile("" : "+v"(w[0]), "+v"(w[1]));
    }
    if (w[0] + w[1] == 12345.678f) sink[0] = w[0];
}

struct Result { double ms_queue, ms_base, batches, records, empty_polls, cas_retries, spins, guard; double sec[5]; };

template <int BODY>
Result run(int waves_per_wg, int cus) {
    unsigned long long* dc; float* sink;
    hipMalloc(&dc, 11 * sizeof(unsigned long long)); hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)queue_kernel<BODY, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds));
    hipFuncSetAttribute((const void*)queue_kernel<BODY, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds));
    hipFuncSetAttribute((const void*)baseline_kernel<BODY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 64 * waves_per_wg;
    Result r{1e30, 1e30, 0, 0, 0, 0, 0, 0, {0, 0, 0, 0, 0}};
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(dc, 0, 11 * sizeof(unsigned long long));
        hipEventRecord(e0);
        queue_kernel<BODY, false><<<cus, threads, sizeof(Lds)>>>(dc, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < r.ms_queue) {
            r.ms_queue = ms;
            unsigned long long c[6]; hipMemcpy(c, dc, sizeof c, hipMemcpyDeviceToHost);
            r.guard = (double)c[5];
            r.batches = (double)c[0] / cus; r.records = (double)c[1] / cus; r.empty_polls = (double)c[2] / cus; r.cas_retries = (double)c[3] / cus; r.spins = (double)c[4] / cus;
        }
    }
    {   // the same kernel with s_memtime stamps around its sections: where a wave's time per batch goes (the stamps cost a little: shares)
        hipMemset(dc, 0, 11 * sizeof(unsigned long long));
        queue_kernel<BODY, true><<<cus, threads, sizeof(Lds)>>>(dc, sink);
        hipDeviceSynchronize();
        unsigned long long c[11]; hipMemcpy(c, dc, sizeof c, hipMemcpyDeviceToHost);
        for (int k = 0; k < 5; ++k) r.sec[k] = c[0] ? (double)c[6 + k] / (double)c[0] : 0.0;
    }
    const int per_wave = NREC * TRIPS / 64 / waves_per_wg;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        baseline_kernel<BODY><<<cus, threads, sizeof(Lds)>>>(per_wave, sink);      // the same LDS request: one workgroup per CU here too
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < r.ms_base) r.ms_base = ms;
    }
    hipFree(dc); hipFree(sink);
    return r;
}

template <int BODY>
void report(int waves_per_wg, int cus, double hz, bool first) {
    const Result r = run<BODY>(waves_per_wg, cus);
    const double full_batches = (double)NREC * TRIPS / 64;                     // per workgroup, if every batch were full
    // SIMD cycles per batch: the workgroup's time x clock x 4 SIMDs / batches
    const double cyc_queue = r.ms_queue * 1e-3 * hz * 4 / r.batches, cyc_base = r.ms_base * 1e-3 * hz * 4 / full_batches;
    // per FULL batch's worth of records (64 paths through one phase): what the design pays per 64 path-phases
    const double cyc_queue_per64 = r.ms_queue * 1e-3 * hz * 4 / (r.records / 64.0);
    printf("%s{\"body_instructions\": %d, \"waves_per_simd\": %d, \"ms_queue\": %.4f, \"ms_body_only\": %.4f, \"batches_per_workgroup\": %.0f, \"mean_batch_fill\": %.2f, "
           "\"simd_cycles_per_batch\": %.1f, \"simd_cycles_per_64_path_phases\": %.1f, \"simd_cycles_body_only_per_batch\": %.1f, "
           "\"queue_overhead_cycles_per_64_path_phases\": %.1f, \"queue_overhead_instruction_equivalents\": %.1f, \"overhead_over_body\": %.3f, "
           "\"empty_polls_per_workgroup\": %.0f, \"cas_retries_per_workgroup\": %.0f, \"valid_bit_spins_per_workgroup\": %.0f, \"waves_stopped_by_guard\": %.0f, "
           "\"wave_cycles_per_batch_by_section\": {\"pick_and_claim\": %.0f, \"indices_and_records_in\": %.0f, \"body\": %.0f, \"records_out\": %.0f, \"push\": %.0f}}",
           first ? "" : ", ", BODY, waves_per_wg / 4, r.ms_queue, r.ms_base, r.batches, r.records / r.batches, cyc_queue, cyc_queue_per64, cyc_base,
           cyc_queue_per64 - cyc_base, (cyc_queue_per64 - cyc_base) / 3.87, (cyc_queue_per64 - cyc_base) / cyc_base, r.empty_polls, r.cas_retries, r.spins, r.guard, r.sec[0], r.sec[1], r.sec[2], r.sec[3], r.sec[4]);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double hz = p.clockRate * 1e3;
    printf("{\"device\": \"%s\", \"cus\": %d, \"nominal_clock_mhz\": %d, \"records_per_workgroup\": %d, \"record_bytes\": %d, \"phase_queues\": %d, \"batches_per_path\": %d, "
           "\"lds_bytes_per_workgroup\": %d, \"rule\": \"GO if at 3 waves per SIMD the queue overhead per 64 path-phases is <= 25 %% of the 100-instruction body (<= 25 instruction-equivalents) "
           "and the mean batch fill is >= 48; fixed before the first run\", \"cases\": [",
           p.gcnArchName, cus, p.clockRate / 1000, NREC, WORDS * 4, NQ, TRIPS, (int)sizeof(Lds));
    bool first = true;
    for (int wpw : {8, 12, 16}) {
        report<40>(wpw, cus, hz, first); first = false;
        report<100>(wpw, cus, hz, false);
        report<200>(wpw, cus, hz, false);
    }
    printf("]}\n");
    return 0;
}
