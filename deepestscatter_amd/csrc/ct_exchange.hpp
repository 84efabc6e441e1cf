// ct_exchange.hpp -- the DELTA estimator with a BLOCK-WIDE EXCHANGE OF PATHS between the waves of a workgroup.
// Included by ct_kernels.hip (inside namespace ct, after render_delta_kernel, whose building blocks it uses).
//
// render_delta_kernel keeps a path in one lane from its first flight to its end, so a wave's lanes are split between the
// two phases of a path -- a tracking visit runs with 43 of 64 lanes on average, a scatter phase with 41 --
// and the kernel, which is bound by instruction issue, pays every instruction for 64.  Here a path changes lanes at the
// two points of its life where its state is small:
//
//     tracking visit --(real collision: position, seed)--> BOUNCE ring --> scatter batch (NEE + new direction)
//          ^                                                                        |
//          +---- dda_begin <-- TRACK ring <--(position, direction, seed, radiance so far)
//
// One workgroup of 1024 threads per CU (16 waves; all 160 KiB of LDS: the tables render_delta_kernel keeps per block, once,
// and the pool).  Every path of the block owns one 48-byte SLOT of the pool for its whole life:
//     A = (pos.xyz, seed)   B = (dir.xyz, depth | flags)   C = (radiance.xyz, result index)
// A lane that tracks a path holds the slot's id plus the flight's state (direction, seed, DDA) in registers and touches the
// slot twice: when it adopts the path (reads A, B) and when the flight ends -- in a real collision (writes A, hands the id
// to the BOUNCE ring) or by leaving the volume (reads C, writes the sample).  Any wave that finds 64 ids in the BOUNCE ring
// takes them, runs the scatter phase for those 64 paths -- a full wave, whatever its own lanes are doing: their flight state
// simply stays in its registers -- writes the slots back and hands the ids to the TRACK ring, from which the lanes that have
// lost their path refill before the next tracking visit.  A finished path leaves its slot to the next sample.
//
// The rings hold 16-bit entries (slot id | lap parity << 15) between two monotone counters.  A producer reserves room with
// ONE atomic add on the tail and then writes its entries; a consumer moves the head with a compare-and-swap bounded by the
// tail it has read, and a lane that finds an entry of the previous lap waits for its writer (which is past its add and
// cannot block).  A ring cannot overflow: it has at least as many entries as the pool has slots, and a slot's id is in at
// most one place.
//
// Nothing here changes a value: a path's arithmetic is the same sequence of operations on the same numbers wherever it
// runs, samples are written to the same place, and the counters are sums.  Parity with the oracle twin is bit for bit, as
// for render_delta_kernel (tests: every DELTA test runs both kernels).

constexpr int kXThreads = 1024;
constexpr int kXWaves = kXThreads / 64;
constexpr int kXRingLog = 11;
constexpr uint32_t kXRing = 1u << kXRingLog;        // entries per ring (>= slots beyond the resident ones, and >= what can be queued)
constexpr uint32_t kXEmpty = 0x80000000u;           // B.w: the slot holds no path -- whoever takes it starts a new sample
constexpr uint32_t kXNoSlot = 0xffffffffu;
constexpr uint32_t kXSpinLimit = 1u << 22;         // watchdog: no wait inside the kernel is unbounded (a fault must not hang the GPU)
constexpr unsigned long long kXWatchdogTicks = 800000000ull;    // 8 s of the 100 MHz wall clock (a launch is well under a second)

struct XCtl {
    uint32_t b_tail, b_head;   // BOUNCE ring: collided paths waiting for a scatter batch
    uint32_t t_tail, t_head;   // TRACK ring: scattered paths waiting for a tracking lane
    uint32_t f_tail, f_head;   // FREE ring: slots without a path, for the lanes that start new samples
    int32_t live;              // paths of this block that have started and not ended
    uint32_t drained_waves;    // waves that will not start another sample
    uint32_t no_jobs;          // some wave has found every job queue empty (then it is empty for all)
    uint32_t pad[3];
};

#define X_FENCE_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#define X_FENCE_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")

CT_DEV uint32_t x_load(const uint32_t *p)
{
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}

// One look at the block's control words per scheduler visit: nine LDS reads in flight together instead of one
// round trip per decision.  The values are hints (a ring is re-read by the pop that acts on it); readfirstlane makes
// the decisions scalar branches.
struct XSnap {
    uint32_t b_pending, t_pending, f_pending, drained_waves, no_jobs;
    int32_t live;
};

CT_DEV XSnap x_snapshot(const XCtl *c)
{
    const uint32_t bt = __hip_atomic_load(&c->b_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), bh = __hip_atomic_load(&c->b_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t tt = __hip_atomic_load(&c->t_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), th = __hip_atomic_load(&c->t_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t ft = __hip_atomic_load(&c->f_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), fh = __hip_atomic_load(&c->f_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t dw = __hip_atomic_load(&c->drained_waves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t nj = __hip_atomic_load(&c->no_jobs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int32_t lv = __hip_atomic_load(&c->live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    XSnap s;
    s.b_pending = __builtin_amdgcn_readfirstlane(bt - bh);
    s.t_pending = __builtin_amdgcn_readfirstlane(tt - th);
    s.f_pending = __builtin_amdgcn_readfirstlane(ft - fh);
    s.drained_waves = __builtin_amdgcn_readfirstlane(dw);
    s.no_jobs = __builtin_amdgcn_readfirstlane(nj);
    s.live = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)lv);
    return s;
}

// The lanes with `mine` append their slot ids to a ring.  (Their slot writes come first: release.)
CT_DEV void x_push(uint16_t *ring, uint32_t *tail, bool mine, uint32_t id, uint32_t lane)
{
    const uint64_t mask = __builtin_amdgcn_ballot_w64(mine);
    if (mask == 0ull) {
        return;
    }
    const uint32_t leader = (uint32_t)__builtin_ctzll(mask);
    uint32_t base = 0;
    if (lane == leader) {
        base = __hip_atomic_fetch_add(tail, (uint32_t)__builtin_popcountll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    base = __builtin_amdgcn_readlane(base, leader);
    X_FENCE_RELEASE();
    if (mine) {
        const uint32_t p = base + lane_rank(mask);
        const uint16_t e = (uint16_t)(id | (((p >> kXRingLog) & 1u) << 15));
        __hip_atomic_store(ring + (p & (kXRing - 1u)), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// Up to popcount(needy) ids from a ring, one for each of the first lanes of `needy` (by rank).  Returns how many.
CT_DEV uint32_t x_pop(uint16_t *ring, uint32_t *head, const uint32_t *tail, uint64_t needy, uint32_t lane, uint32_t &id, bool &got, bool &bad)
{
    got = false;
    const uint32_t want = (uint32_t)__builtin_popcountll(needy);
    uint32_t n = 0, base = 0;
    if (lane == 0) {
        uint32_t h = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (;;) {
            const uint32_t t = __hip_atomic_load(tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t take = min(t - h, want);
            if (take == 0u) {
                break;
            }
            uint32_t expected = h;
            if (__hip_atomic_compare_exchange_strong(head, &expected, h + take, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                n = take;
                base = h;
                break;
            }
            h = expected;
        }
    }
    n = __builtin_amdgcn_readfirstlane(n);
    base = __builtin_amdgcn_readfirstlane(base);
    if (n == 0u) {
        return 0u;
    }
    const uint32_t rank = lane_rank(needy);
    if (((needy >> lane) & 1ull) != 0ull && rank < n) {
        const uint32_t p = base + rank, tag = (p >> kXRingLog) & 1u;
        uint32_t e = __hip_atomic_load(ring + (p & (kXRing - 1u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t spins = 0;
        while ((e >> 15) != tag && spins < kXSpinLimit) {   // reserved by a producer that is about to write it
            __builtin_amdgcn_s_sleep(1);
            e = __hip_atomic_load(ring + (p & (kXRing - 1u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            spins += 1;
        }
        id = e & 0x7fffu;
        got = (e >> 15) == tag;   // (false only if the watchdog bound was hit: the caller reports it)
        bad = bad || !got;
    }
    X_FENCE_ACQUIRE();
    return n;
}

template <int MODE, bool STATS>
__global__ __launch_bounds__(kXThreads) void render_delta_x_kernel(DevScene sc, BatchArgs ba, uint32_t n_slots, uint32_t n_scatter_waves)
{
    __shared__ MieLds lds;
    __shared__ uint32_t maj_words[kMajCellsMax / 4];
    __shared__ uint8_t lds_codes[kMajCellsMax / 4];
    __shared__ float2 sigma_table[256];
    __shared__ XCtl ctl;
    extern __shared__ uint4 x_pool[];
    uint4 *const slotA = x_pool, *const slotB = slotA + n_slots, *const slotC = slotB + n_slots;
    uint16_t *const ringB = (uint16_t *)(slotC + n_slots), *const ringT = ringB + kXRing, *const ringF = ringT + kXRing;
    {
        const uint32_t words = ((uint32_t)(sc.mc_gx * sc.mc_gy * sc.mc_gz) + 3u) >> 2;
        const uint32_t *src = (const uint32_t *)sc.maj_cells, *srcc = (const uint32_t *)sc.maj_codes;
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) {
            maj_words[i] = src[i];
            const uint32_t w = srcc[i];
            lds_codes[i] = (uint8_t)((w & 3u) | ((w >> 6) & 0xcu) | ((w >> 12) & 0x30u) | ((w >> 18) & 0xc0u));
        }
        if (threadIdx.x < 256u) {
            const float sb = ((float)threadIdx.x * (1.0f / 255.0f)) * sc.density_multiplier;
            sigma_table[threadIdx.x] = make_float2(sb, 1.0f / sb);
        }
        // the slots beyond the resident ones start in the FREE ring (lap 0: parity 0); every other ring entry carries the
        // parity of "lap -1"
        const uint32_t spare = n_slots - (uint32_t)kXThreads;
        for (uint32_t i = threadIdx.x; i < kXRing; i += blockDim.x) {
            ringB[i] = 0x8000u;
            ringT[i] = 0x8000u;
            ringF[i] = (i < spare) ? (uint16_t)((uint32_t)kXThreads + i) : (uint16_t)0x8000u;
        }
        if (threadIdx.x == 0) {
            ctl.b_tail = ctl.b_head = 0u;
            ctl.t_tail = ctl.t_head = 0u;
            ctl.f_head = 0u;
            ctl.f_tail = spare;
            ctl.live = 0;
            ctl.drained_waves = n_scatter_waves;   // (a scatter wave never starts a sample)
            ctl.no_jobs = 0u;
        }
    }
    const uint8_t *lds_maj = (const uint8_t *)maj_words;
    load_tables(sc, lds);   // (ends with the block's barrier)

    const uint32_t lane = threadIdx.x & 63u;
    // this lane's flight (state == ST_MARCH), and the slot it holds (with a path in flight, or empty, or none)
    f3 dir = mk3(0, 0, 1);
    Dda dda{};
    uint32_t seed = 0, slot = threadIdx.x;
    int state = ST_IDLE;

    uint32_t q_next = 0, q_end = 0, job_g = 0, job_s0 = 0;
    uint32_t q_cur = (uint32_t)kQueues, q_tried = 0;
    bool drained = false, announced = false;
    uint32_t c_dl = 0, c_il = 0, c_cap = 0;
    uint32_t st_march = 0, st_march_l = 0, st_scat = 0, st_scat_l = 0, st_regen = 0, st_regen_l = 0, st_adopt = 0, st_adopt_l = 0, st_spin = 0;
    uint32_t iv_dealt = 0, iv_written = 0;
    int32_t ended_paths = 0;     // path ends this wave has not yet taken off ctl.live (wave-uniform)
    bool bad = false;            // a bounded wait ran out (watchdog; reported in stats[63], tests assert 0)
    const unsigned long long t_start = wall_clock64();
    uint32_t visit = 0;

    // The scatter phase of render_delta_kernel (cloudRadianceMaterials.cu:53-61) for up to 64 paths of the BOUNCE ring.
    auto scatter_batch = [&](const XSnap &snap) {
        uint32_t id = 0;
        bool got;
        const uint32_t n = x_pop(ringB, &ctl.b_head, &ctl.b_tail, ~0ull, lane, id, got, bad);
        if (n == 0u) {
            return;
        }
        bool onward = false;
        if (STATS) {
            st_scat += 1;
            st_scat_l += n;
        }
        if (got) {
            const uint4 a = slotA[id], b = slotB[id], c = slotC[id];
            const f3 pos = mk3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
            f3 d = mk3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
            f3 rad = mk3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z));
            uint32_t s = a.w, depth = b.w;
            const bool chopped = (MODE == 1) ? true : (MODE == 0 ? (depth != 1u) : false);
            const NeeLoads nee = in_scattering_issue(sc, pos, d, chopped);
            c_il += 1;
            bool go = (MODE != 2);
            if (go) {
                d = new_direction(lds.cdf, lds.guide, s, d);
                depth++;
                if (depth == sc.max_depth) {
                    c_cap += 1;
                    go = false;
                }
            }
            rad = add3(rad, in_scattering_finish(sc, nee, pos));
            if (go) {
                slotA[id].w = s;
                slotB[id] = make_uint4(__float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z), depth);
                slotC[id] = make_uint4(__float_as_uint(rad.x), __float_as_uint(rad.y), __float_as_uint(rad.z), c.w);
                onward = true;
            } else {
                ba.frames[c.w] = make_float4(rad.x, rad.y, rad.z, 1.f);
                if (STATS) {
                    iv_written += 1;
                }
            }
        }
        const uint32_t n_onward = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(onward));
        ended_paths += (int32_t)(n - n_onward);
        x_push(ringT, &ctl.t_tail, onward, id, lane);
        // a finished path's slot is free for a new sample while some wave may still start one
        if (n_onward != n && snap.drained_waves != (uint32_t)kXWaves) {
            x_push(ringF, &ctl.f_tail, got && !onward, id, lane);
        }
    };
    auto flush_ended = [&]() {
        if (ended_paths != 0) {
            if (lane == 0) {
                __hip_atomic_fetch_add(&ctl.live, -ended_paths, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            ended_paths = 0;
        }
    };

    if ((threadIdx.x >> 6) < n_scatter_waves) {
        // ---------------- a SCATTER wave: batches of 64 collided paths, nothing else ----------------
        for (;;) {
            visit += 1;
            if ((visit & 1023u) == 0u && wall_clock64() - t_start > kXWatchdogTicks) {
                bad = true;
            }
            if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
                break;
            }
            const XSnap snap = x_snapshot(&ctl);
            // a full batch; a partial one when the tracking lanes have nothing to adopt (the supply is short, or it is the tail)
            if (snap.b_pending >= 64u || (snap.b_pending != 0u && snap.t_pending == 0u)) {
                scatter_batch(snap);
                flush_ended();
            } else if (snap.drained_waves == (uint32_t)kXWaves && snap.live == 0) {
                break;
            } else {
                __builtin_amdgcn_s_sleep(2);
                if (STATS) {
                    st_spin += 1;
                }
            }
        }
    } else
    for (;;) {
        visit += 1;
        if ((visit & 1023u) == 0u && wall_clock64() - t_start > kXWatchdogTicks) {
            bad = true;
        }
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
            break;   // give up (results are wrong): the grid must drain whatever happened
        }
        const XSnap snap = x_snapshot(&ctl);
        // ---------------- (1) scatter duty of a TRACKING wave ----------------
        {
            bool duty;
            if (n_scatter_waves == 0u) {
                // every wave does both: a full batch, or what is left once this wave has nothing else to do
                duty = snap.b_pending >= 64u ||
                       (snap.b_pending != 0u && drained && q_next == q_end && __builtin_amdgcn_ballot_w64(state == ST_MARCH) == 0ull);
            } else {
                duty = snap.b_pending >= 256u;   // the scatter waves are behind
            }
            if (duty) {
                scatter_batch(snap);
            }
        }

        // ---------------- (2) refill: lanes without a path adopt one from the TRACK ring ----------------
        if (!drained && q_next == q_end && snap.no_jobs != 0u) {
            drained = true;   // (nothing left to take, and this wave's own job is used up)
        }
        {
            const uint64_t needy = __builtin_amdgcn_ballot_w64(state == ST_IDLE && slot == kXNoSlot);
            if (needy != 0ull && (snap.t_pending != 0u || snap.b_pending >= 64u)) {   // (a batch of step 1 may just have added some)
                uint32_t id = 0;
                bool got;
                const uint32_t n = x_pop(ringT, &ctl.t_head, &ctl.t_tail, needy, lane, id, got, bad);
                if (STATS && n != 0u) {
                    st_adopt += 1;
                    st_adopt_l += n;
                }
                if (got) {
                    slot = id;
                    const uint4 a = slotA[id], b = slotB[id];
                    dir = mk3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
                    seed = a.w;
                    dda_begin(sc, dda, mk3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)), dir);
                    state = ST_MARCH;
                }
            }
        }
        // ---------------- (2b) new samples (as render_delta_kernel regenerates), in free slots ----------------
        {
            const bool may_start = !(drained && q_next == q_end);
            // lanes that still have nothing take a free slot, if this wave has samples to start
            const uint64_t slotless = __builtin_amdgcn_ballot_w64(state == ST_IDLE && slot == kXNoSlot);
            if (may_start && slotless != 0ull && snap.f_pending != 0u) {
                uint32_t id = 0;
                bool got;
                x_pop(ringF, &ctl.f_head, &ctl.f_tail, slotless, lane, id, got, bad);
                if (got) {
                    slot = id;
                }
            }
            const bool empty = state == ST_IDLE && slot != kXNoSlot;
            const uint64_t idle = __builtin_amdgcn_ballot_w64(empty);
            const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
            if (n_idle != 0u && !may_start) {
                // nothing left to start here: the slots go to the waves that still start samples, or out of circulation
                if (snap.drained_waves != (uint32_t)kXWaves) {
                    x_push(ringF, &ctl.f_tail, empty, slot, lane);
                }
                if (empty) {
                    slot = kXNoSlot;
                }
            } else if (n_idle >= sc.regen_min || n_idle == 64u || (n_idle != 0u && __builtin_amdgcn_ballot_w64(state == ST_MARCH) == 0ull)) {
                if (q_next == q_end) {
                    uint32_t j = 0;
                    if (!take_job(ba, lane, q_cur, q_tried, j)) {
                        drained = true;
                        if (lane == 0) {
                            __hip_atomic_store(&ctl.no_jobs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    } else {
                        const uint32_t sub = __builtin_amdgcn_readfirstlane(ba.job_sub[j]);
                        job_g = __builtin_amdgcn_readfirstlane(ba.job_group[j]);
                        job_s0 = sub & 0xffffu;
                        q_next = 0;
                        q_end = (job_s0 < ba.S ? min(sub >> 16, ba.S - job_s0) : 0u) * 64u;
                    }
                }
                if (q_next != q_end) {
                    const uint32_t avail = q_end - q_next;
                    if (STATS) {
                        st_regen += 1;
                        st_regen_l += min(n_idle, avail);
                    }
                    const uint32_t rank = lane_rank(idle);
                    const bool take = empty && rank < avail;
                    const uint32_t q = q_next + rank;
                    q_next += min(n_idle, avail);
                    bool started = false;
                    if (take) {
                        const uint32_t s = job_s0 + (q >> 6), l = q & 63u;
                        const uint32_t g = job_g;
                        const uint32_t pixel = ba.pixels[g * 64u + l];
                        if (pixel != 0xffffffffu) {
                            if (STATS) {
                                iv_dealt += 1;
                            }
                            const float4 p0 = ba.primary[2 * (size_t)pixel];
                            const float4 p1 = ba.primary[2 * (size_t)pixel + 1];
                            const uint32_t out_idx = ba.frame_stride ? ba.out_offset + s * ba.frame_stride + group_column(ba, g) + l : pixel;
                            const f3 pos = mk3(p0.x, p0.y, p0.z);
                            const bool hit = p0.w != 0.f;
                            dir = mk3(p1.x, p1.y, p1.z);
                            seed = tea4(__float_as_uint(p1.w), ba.first_subframe + s);
                            uint32_t depth = 0;
                            if (MODE == 1) {
                                dir = new_direction(lds.cdf, lds.guide, seed, dir);
                            }
                            bool go = hit && in_box(sc, pos);
                            if (MODE != 2 && go) {
                                depth = 1;
                                if (depth == sc.max_depth) {
                                    c_cap += 1;
                                    go = false;
                                }
                            }
                            if (go) {
                                if (MODE != 1 && ba.advance) {
                                    const float4 a0 = ba.advance[4 * (size_t)pixel], a1 = ba.advance[4 * (size_t)pixel + 1];
                                    const float4 a2 = ba.advance[4 * (size_t)pixel + 2], a3 = ba.advance[4 * (size_t)pixel + 3];
                                    dda.org = mk3(a0.x, a0.y, a0.z);
                                    dda.t = a0.w;
                                    dda.tmax = mk3(a1.x, a1.y, a1.z);
                                    dda.bx = __float_as_int(a1.w);
                                    dda.tdelta = mk3(a2.x, a2.y, a2.z);
                                    dda.by = __float_as_int(a2.w);
                                    dda.bz = __float_as_int(a3.x);
                                } else {
                                    dda_begin(sc, dda, pos, dir);
                                }
                                slotB[slot] = make_uint4(__float_as_uint(dir.x), __float_as_uint(dir.y), __float_as_uint(dir.z), depth);
                                slotC[slot] = make_uint4(0u, 0u, 0u, out_idx);   // radiance (0, 0, 0)
                                state = ST_MARCH;
                                started = true;
                            } else {
                                ba.frames[out_idx] = make_float4(0.f, 0.f, 0.f, 1.f);
                                if (STATS) {
                                    iv_written += 1;
                                }
                            }
                        }
                    }
                    // the block's count of paths in flight rises BEFORE any of them can reach another wave
                    const uint64_t st = __builtin_amdgcn_ballot_w64(started);
                    if (st != 0ull && lane == (uint32_t)__builtin_ctzll(st)) {
                        __hip_atomic_fetch_add(&ctl.live, (int32_t)__builtin_popcountll(st), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }

        // ---------------- (3) tracking visits (render_delta_kernel's, for the lanes in flight) ----------------
        bool collided = false;
        {
            uint32_t burst = sc.march_burst;
            for (;;) {
                const uint64_t marching = __builtin_amdgcn_ballot_w64(state == ST_MARCH);
                if (marching == 0ull) {
                    break;
                }
                if (STATS) {
                    st_march += 1;
                    st_march_l += (uint32_t)__builtin_popcountll(marching);
                }
                bool ended = false;
                if (state == ST_MARCH) {
                    bool collide = false;
                    float sigma_bar = 0.0f, sigma_low = 0.0f;
                    if (!cell_in_grid(sc, dda)) {
                        ended = true;
                    } else {
                        const uint32_t ci = cell_index(sc, dda);
                        const uint32_t M = lds_maj[ci];
                        if (M != 0u) {
                            const uint32_t q = ((uint32_t)lds_codes[ci >> 2] >> ((ci & 3u) * 2u)) & 3u;
                            sigma_low = sigma_table[(q * M) >> 2].x;
                            const float2 sb = sigma_table[M];
                            sigma_bar = sb.x;
                            const float u = u24_to_float(lcg24(seed));
                            const float dt = -logf_above_one(1.0f - u) * sb.y;
                            const float t_exit = fminf(fminf(dda.tmax.x, dda.tmax.y), dda.tmax.z);
                            if (dda.t + dt < t_exit) {
                                dda.t = dda.t + dt;
                                collide = true;
                            }
                        }
                        if (!collide) {
                            dda_cross(dda, dir);
                        }
                    }
                    if (collide) {
                        const f3 p = mk3(fmaf(dir.x, dda.t, dda.org.x), fmaf(dir.y, dda.t, dda.org.y), fmaf(dir.z, dda.t, dda.org.z));
                        const float z = u24_to_float(lcg24(seed));
                        bool real = z * sigma_bar < sigma_low;
                        if (!real) {
                            uint32_t meta_unused;
                            const uint2 cell = fetch_cell_in_grid(sc, sc.dbricks, p, meta_unused);
                            c_dl += 1;
                            real = z * sigma_bar < filter_at(sc, cell, p) * sc.density_multiplier;
                        }
                        if (real) {
                            if (in_box(sc, p)) {
                                slotA[slot] = make_uint4(__float_as_uint(p.x), __float_as_uint(p.y), __float_as_uint(p.z), seed);
                                collided = true;
                                state = ST_IDLE;
                            } else {
                                ended = true;
                            }
                        }
                    }
                    if (ended) {
                        const uint4 c = slotC[slot];
                        ba.frames[c.w] = make_float4(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), 1.f);
                        if (STATS) {
                            iv_written += 1;
                        }
                        state = ST_IDLE;   // (the slot stays with the lane, empty)
                    }
                }
                ended_paths += (int32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ended));   // (wave-uniform: outside the branch)
                if (--burst == 0u) {
                    break;
                }
            }
        }
        // the collided paths go to the BOUNCE ring; their lanes are free for another path
        x_push(ringB, &ctl.b_tail, collided, slot, lane);
        if (collided) {
            slot = kXNoSlot;
        }

        // ---------------- (4) bookkeeping, exit ----------------
        flush_ended();
        if (drained && q_next == q_end) {
            if (!announced) {
                announced = true;
                if (lane == 0) {
                    __hip_atomic_fetch_add(&ctl.drained_waves, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (__builtin_amdgcn_ballot_w64(state == ST_MARCH) == 0ull) {
                // (the snapshot is from before this visit: a wave leaves one visit after the last path has gone)
                if (snap.drained_waves == (uint32_t)kXWaves && snap.live == 0) {
                    break;
                }
                if (snap.b_pending == 0u && snap.t_pending == 0u) {
                    __builtin_amdgcn_s_sleep(8);   // the last paths are in other waves' lanes
                    if (STATS) {
                        st_spin += 1;
                    }
                }
            }
        }
    }

    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) {
        atomicAdd(&ba.stats[63], 1ull);
    }
    uint32_t vals[3] = { c_dl, c_il, c_cap };
#pragma unroll
    for (int i = 0; i < 3; i++) {
        uint32_t v = vals[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v += __shfl_xor(v, off);
        }
        vals[i] = v;
    }
    if (STATS) {
        uint32_t sv[2] = { iv_dealt, iv_written };
#pragma unroll
        for (int i = 0; i < 2; i++) {
            uint32_t v = sv[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                v += __shfl_xor(v, off);
            }
            sv[i] = v;
        }
        if (lane == 0) {
            atomicAdd(&ba.stats[0], (unsigned long long)st_regen);
            atomicAdd(&ba.stats[1], (unsigned long long)st_regen_l);
            atomicAdd(&ba.stats[2], (unsigned long long)st_march);
            atomicAdd(&ba.stats[3], (unsigned long long)st_march_l);
            atomicAdd(&ba.stats[4], (unsigned long long)st_scat);
            atomicAdd(&ba.stats[5], (unsigned long long)st_scat_l);
            atomicAdd(&ba.stats[33], (unsigned long long)st_adopt);
            atomicAdd(&ba.stats[34], (unsigned long long)st_adopt_l);
            atomicAdd(&ba.stats[35], (unsigned long long)st_spin);
            atomicAdd(&ba.stats[64], (unsigned long long)sv[0]);
            atomicAdd(&ba.stats[66], (unsigned long long)sv[1]);
        }
    }
    if (lane == 0) {
        atomicAdd(&ba.counters[2], (unsigned long long)vals[0]);
        atomicAdd(&ba.counters[3], (unsigned long long)vals[1]);
        atomicAdd(&ba.counters[4], (unsigned long long)vals[1]);
        atomicAdd(&ba.counters[5], (unsigned long long)vals[2]);
        atomicAdd(&ba.counters[6], (unsigned long long)vals[0]);
        atomicAdd(&ba.counters[7], (unsigned long long)vals[1]);
    }
}

// Bytes of dynamic LDS for a pool of n_slots slots + the two rings.
inline size_t delta_x_pool_bytes(uint32_t n_slots)
{
    return (size_t)n_slots * 48u + 3u * kXRing * sizeof(uint16_t);
}

hipError_t launch_render_delta_x(const DevScene &sc, const BatchArgs &ba, LaunchShape shape, hipStream_t stream)
{
    const dim3 grid(shape.blocks), block(kXThreads);
    const uint32_t n_slots = shape.pool_slots;
    const size_t dyn = delta_x_pool_bytes(n_slots);
#define CT_X_LAUNCH(M, S) hipLaunchKernelGGL((render_delta_x_kernel<M, S>), grid, block, dyn, stream, sc, ba, n_slots, shape.scatter_waves)
    if (shape.stats) {
        switch (sc.mode) {
        case 0: CT_X_LAUNCH(0, true); break;
        case 1: CT_X_LAUNCH(1, true); break;
        default: CT_X_LAUNCH(2, true); break;
        }
    } else {
        switch (sc.mode) {
        case 0: CT_X_LAUNCH(0, false); break;
        case 1: CT_X_LAUNCH(1, false); break;
        default: CT_X_LAUNCH(2, false); break;
        }
    }
#undef CT_X_LAUNCH
    return hipGetLastError();
}

// One block per CU; the pool takes what the tables leave of the CU's 160 KiB.
LaunchShape exchange_shape(int device)
{
    LaunchShape s{ 256, kXThreads, false };
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
        s.blocks = prop.multiProcessorCount;
    }
    hipFuncAttributes fa{};
    size_t fixed = 80 * 1024;
    if (hipFuncGetAttributes(&fa, (const void *)render_delta_x_kernel<0, false>) == hipSuccess) {
        fixed = fa.sharedSizeBytes;
    }
    const size_t room = 160 * 1024 - fixed - 3u * kXRing * sizeof(uint16_t);
    uint32_t slots = (uint32_t)(room / 48u);
    slots = std::min(slots & ~1u, (uint32_t)kXThreads + kXRing);
    if (const char *e = getenv("CT_X_SLOTS")) {
        const int v = atoi(e);
        if (v >= kXThreads && (uint32_t)v <= slots) {
            slots = (uint32_t)v & ~1u;
        }
    }
    // more than 64 KiB of LDS per block in all: every instantiation is told how much dynamic LDS it may be launched with
    const int dyn = (int)delta_x_pool_bytes(slots);
    (void)hipFuncSetAttribute((const void *)render_delta_x_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    (void)hipFuncSetAttribute((const void *)render_delta_x_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    (void)hipFuncSetAttribute((const void *)render_delta_x_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    (void)hipFuncSetAttribute((const void *)render_delta_x_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    (void)hipFuncSetAttribute((const void *)render_delta_x_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    (void)hipFuncSetAttribute((const void *)render_delta_x_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    (void)hipGetLastError();
    s.pool_slots = slots;
    s.scatter_waves = 6;   // of the block's 16 (CT_X_SCATTER_WAVES; 0 = every wave does both)
    if (const char *e = getenv("CT_X_SCATTER_WAVES")) {
        s.scatter_waves = (uint32_t)std::min(kXWaves - 1, std::max(0, atoi(e)));
    }
    return s;
}

// =============================================================================================================================
// render_delta_w_kernel: the same regrouping of paths by phase, but WITHIN a wave -- no atomics, no fences, no other wave.
//
// The block-wide exchange above pays for its generality: per scheduler visit a snapshot of the control words, a compare-and-swap
// per pop, an atomic add per push, three rings -- +25 % instructions in all instead of fewer, and lanes that wait for whichever
// wave comes round to the scatter batch (profiles/r03c).  A wave does not need the other waves to fill its lanes; it needs more
// paths than lanes.  Here every wave owns S slots of the block's pool (S = 112 with 16 waves and the 84 KiB the tables leave) and
// three private lists of slot indices, kept as stacks whose heights live in SGPRs:
//     BOUNCE  paths that collided and wait for the scatter phase      TRACK  scattered paths that wait for a lane
//     FREE    slots without a path
// A tracking visit's collided lanes push their slot on BOUNCE (position = height + the lane's rank in the ballot) and adopt the
// top entries of TRACK; when BOUNCE holds 64 entries the wave runs the scatter phase for them on all 64 lanes -- whatever those
// lanes hold as tracking state stays in their registers -- and pushes them on TRACK.  Everything is wave-synchronous: LDS executes a wave's
// instructions in order, so a list entry written by one lane is read by another without any synchronisation instruction.
// Same slots (A, B, C), same arithmetic per path, same results as render_delta_kernel bit for bit.
// =============================================================================================================================
constexpr uint32_t kWListCap = 128;   // entries per private list (>= slots per wave)

template <int MODE, bool STATS>
__global__ __launch_bounds__(kXThreads) void render_delta_w_kernel(DevScene sc, BatchArgs ba, uint32_t slots_per_wave)
{
    __shared__ MieLds lds;
    __shared__ uint32_t maj_words[kMajCellsMax / 4];
    __shared__ uint8_t lds_codes[kMajCellsMax / 4];
    __shared__ float2 sigma_table[256];
    extern __shared__ uint4 x_pool[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_slots = slots_per_wave * (uint32_t)kXWaves;
    uint4 *const slotA = x_pool + wave * slots_per_wave, *const slotB = slotA + n_slots, *const slotC = slotB + n_slots;
    uint8_t *const lists = (uint8_t *)(x_pool + 3u * n_slots) + wave * 3u * kWListCap;
    uint8_t *const listB = lists, *const listT = lists + kWListCap, *const listF = lists + 2u * kWListCap;
    {
        const uint32_t words = ((uint32_t)(sc.mc_gx * sc.mc_gy * sc.mc_gz) + 3u) >> 2;
        const uint32_t *src = (const uint32_t *)sc.maj_cells, *srcc = (const uint32_t *)sc.maj_codes;
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) {
            maj_words[i] = src[i];
            const uint32_t w = srcc[i];
            lds_codes[i] = (uint8_t)((w & 3u) | ((w >> 6) & 0xcu) | ((w >> 12) & 0x30u) | ((w >> 18) & 0xc0u));
        }
        if (threadIdx.x < 256u) {
            const float sb = ((float)threadIdx.x * (1.0f / 255.0f)) * sc.density_multiplier;
            sigma_table[threadIdx.x] = make_float2(sb, 1.0f / sb);
        }
        // the wave's slots beyond its lanes' own start on its FREE list
        for (uint32_t i = lane; i + 64u < slots_per_wave; i += 64u) {
            listF[i] = (uint8_t)(64u + i);
        }
    }
    const uint8_t *lds_maj = (const uint8_t *)maj_words;
    load_tables(sc, lds);   // (ends with the block's barrier)

    uint32_t nb = 0, nt = 0, nf = slots_per_wave - 64u;   // heights of the three lists (wave-uniform)
    f3 dir = mk3(0, 0, 1);
    Dda dda{};
    uint32_t seed = 0, slot = lane;    // this lane's slot (index within the wave's region) or kXNoSlot
    int state = ST_IDLE;

    uint32_t q_next = 0, q_end = 0, job_g = 0, job_s0 = 0;
    uint32_t q_cur = (uint32_t)kQueues, q_tried = 0;
    bool drained = false;
    uint32_t c_dl = 0, c_il = 0, c_cap = 0;
    uint32_t st_march = 0, st_march_l = 0, st_scat = 0, st_scat_l = 0, st_regen = 0, st_regen_l = 0, st_adopt = 0, st_adopt_l = 0;
    uint32_t iv_dealt = 0, iv_written = 0;

    const unsigned long long t_start = wall_clock64();
    uint32_t visit = 0;
    for (;;) {
        visit += 1;
        if ((visit & 1023u) == 0u && wall_clock64() - t_start > kXWatchdogTicks) {
            if (lane == 0) {
                atomicAdd(&ba.stats[63], 1ull);   // watchdog (tests assert 0): give up rather than hang the GPU
            }
            break;
        }
        const uint64_t marching0 = __builtin_amdgcn_ballot_w64(state == ST_MARCH);
        const bool may_start = !(drained && q_next == q_end);
        // ---------------- (1) the scatter phase for up to 64 collided paths ----------------
        // (a partial batch: when the wave has nothing else to do, or when a quarter of its lanes can get neither a queued path
        // nor a new sample while half a batch waits here)
        const uint32_t lanes_free = 64u - (uint32_t)__builtin_popcountll(marching0);
        if (nb >= 64u || (nb != 0u && nt == 0u && (marching0 == 0ull || (nb >= 32u && lanes_free >= 16u && (nf == 0u || !may_start))))) {
            const uint32_t n = min(nb, 64u);
            nb -= n;
            bool onward = false, finished = false;
            uint32_t id = 0;
            if (STATS) {
                st_scat += 1;
                st_scat_l += n;
            }
            if (lane < n) {
                id = listB[nb + lane];
                const uint4 a = slotA[id], b = slotB[id], c = slotC[id];
                const f3 pos = mk3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
                f3 d = mk3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
                f3 rad = mk3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z));
                uint32_t s = a.w, depth = b.w;
                // ---- the scatter phase of render_delta_kernel (cloudRadianceMaterials.cu:53-61) ----
                const bool chopped = (MODE == 1) ? true : (MODE == 0 ? (depth != 1u) : false);
                const NeeLoads nee = in_scattering_issue(sc, pos, d, chopped);
                c_il += 1;
                bool go = (MODE != 2);
                if (go) {
                    d = new_direction(lds.cdf, lds.guide, s, d);
                    depth++;
                    if (depth == sc.max_depth) {
                        c_cap += 1;
                        go = false;
                    }
                }
                rad = add3(rad, in_scattering_finish(sc, nee, pos));
                if (go) {
                    slotA[id].w = s;
                    slotB[id] = make_uint4(__float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z), depth);
                    slotC[id] = make_uint4(__float_as_uint(rad.x), __float_as_uint(rad.y), __float_as_uint(rad.z), c.w);
                    onward = true;
                } else {
                    ba.frames[c.w] = make_float4(rad.x, rad.y, rad.z, 1.f);
                    if (STATS) {
                        iv_written += 1;
                    }
                    finished = true;
                }
            }
            const uint64_t on = __builtin_amdgcn_ballot_w64(onward), fin = __builtin_amdgcn_ballot_w64(finished);
            if (onward) {
                listT[nt + lane_rank(on)] = (uint8_t)id;
            }
            if (finished) {
                listF[nf + lane_rank(fin)] = (uint8_t)id;
            }
            nt += (uint32_t)__builtin_popcountll(on);
            nf += (uint32_t)__builtin_popcountll(fin);
        }

        // ---------------- (2) lanes without a path adopt the top of TRACK ----------------
        {
            const bool want = state == ST_IDLE && slot == kXNoSlot;
            const uint64_t needy = __builtin_amdgcn_ballot_w64(want);
            if (needy != 0ull && nt != 0u) {
                const uint32_t k = min((uint32_t)__builtin_popcountll(needy), nt);
                nt -= k;
                const uint32_t rank = lane_rank(needy);
                if (STATS) {
                    st_adopt += 1;
                    st_adopt_l += k;
                }
                if (want && rank < k) {
                    const uint32_t id = listT[nt + rank];
                    slot = id;
                    const uint4 a = slotA[id], b = slotB[id];
                    dir = mk3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
                    seed = a.w;
                    dda_begin(sc, dda, mk3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)), dir);
                    state = ST_MARCH;
                }
            }
        }
        // ---------------- (2b) new samples in free slots (as render_delta_kernel regenerates) ----------------
        {
            if (may_start) {
                // lanes that still have nothing take a free slot
                const bool want = state == ST_IDLE && slot == kXNoSlot;
                const uint64_t needy = __builtin_amdgcn_ballot_w64(want);
                if (needy != 0ull && nf != 0u) {
                    const uint32_t k = min((uint32_t)__builtin_popcountll(needy), nf);
                    nf -= k;
                    const uint32_t rank = lane_rank(needy);
                    if (want && rank < k) {
                        slot = listF[nf + rank];
                    }
                }
            }
            const bool empty = state == ST_IDLE && slot != kXNoSlot;
            const uint64_t idle = __builtin_amdgcn_ballot_w64(empty);
            const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
            if (n_idle != 0u && !may_start) {
                if (empty) {
                    slot = kXNoSlot;   // nothing left to start: the lane is free for a queued path, the slot goes out of use
                }
            } else if (n_idle >= sc.regen_min || n_idle == 64u || (n_idle != 0u && __builtin_amdgcn_ballot_w64(state == ST_MARCH) == 0ull)) {
                if (q_next == q_end) {
                    uint32_t j = 0;
                    if (!take_job(ba, lane, q_cur, q_tried, j)) {
                        drained = true;
                    } else {
                        const uint32_t sub = __builtin_amdgcn_readfirstlane(ba.job_sub[j]);
                        job_g = __builtin_amdgcn_readfirstlane(ba.job_group[j]);
                        job_s0 = sub & 0xffffu;
                        q_next = 0;
                        q_end = (job_s0 < ba.S ? min(sub >> 16, ba.S - job_s0) : 0u) * 64u;
                    }
                }
                if (q_next != q_end) {
                    const uint32_t avail = q_end - q_next;
                    if (STATS) {
                        st_regen += 1;
                        st_regen_l += min(n_idle, avail);
                    }
                    const uint32_t rank = lane_rank(idle);
                    const bool take = empty && rank < avail;
                    const uint32_t q = q_next + rank;
                    q_next += min(n_idle, avail);
                    if (take) {
                        const uint32_t s = job_s0 + (q >> 6), l = q & 63u;
                        const uint32_t g = job_g;
                        const uint32_t pixel = ba.pixels[g * 64u + l];
                        if (pixel != 0xffffffffu) {
                            if (STATS) {
                                iv_dealt += 1;
                            }
                            const float4 p0 = ba.primary[2 * (size_t)pixel];
                            const float4 p1 = ba.primary[2 * (size_t)pixel + 1];
                            const uint32_t out_idx = ba.frame_stride ? ba.out_offset + s * ba.frame_stride + group_column(ba, g) + l : pixel;
                            const f3 pos = mk3(p0.x, p0.y, p0.z);
                            const bool hit = p0.w != 0.f;
                            dir = mk3(p1.x, p1.y, p1.z);
                            seed = tea4(__float_as_uint(p1.w), ba.first_subframe + s);
                            uint32_t depth = 0;
                            if (MODE == 1) {
                                dir = new_direction(lds.cdf, lds.guide, seed, dir);
                            }
                            bool go = hit && in_box(sc, pos);
                            if (MODE != 2 && go) {
                                depth = 1;
                                if (depth == sc.max_depth) {
                                    c_cap += 1;
                                    go = false;
                                }
                            }
                            if (go) {
                                if (MODE != 1 && ba.advance) {
                                    const float4 a0 = ba.advance[4 * (size_t)pixel], a1 = ba.advance[4 * (size_t)pixel + 1];
                                    const float4 a2 = ba.advance[4 * (size_t)pixel + 2], a3 = ba.advance[4 * (size_t)pixel + 3];
                                    dda.org = mk3(a0.x, a0.y, a0.z);
                                    dda.t = a0.w;
                                    dda.tmax = mk3(a1.x, a1.y, a1.z);
                                    dda.bx = __float_as_int(a1.w);
                                    dda.tdelta = mk3(a2.x, a2.y, a2.z);
                                    dda.by = __float_as_int(a2.w);
                                    dda.bz = __float_as_int(a3.x);
                                } else {
                                    dda_begin(sc, dda, pos, dir);
                                }
                                slotB[slot] = make_uint4(__float_as_uint(dir.x), __float_as_uint(dir.y), __float_as_uint(dir.z), depth);
                                slotC[slot] = make_uint4(0u, 0u, 0u, out_idx);   // radiance (0, 0, 0)
                                state = ST_MARCH;
                            } else {
                                ba.frames[out_idx] = make_float4(0.f, 0.f, 0.f, 1.f);
                                if (STATS) {
                                    iv_written += 1;
                                }
                            }
                        }
                    }
                }
            }
        }

        // ---------------- (3) tracking visits (render_delta_kernel's, for the lanes in flight) ----------------
        bool collided = false;
        {
            uint32_t burst = sc.march_burst;
            for (;;) {
                const uint64_t marching = __builtin_amdgcn_ballot_w64(state == ST_MARCH);
                if (marching == 0ull) {
                    break;
                }
                if (STATS) {
                    st_march += 1;
                    st_march_l += (uint32_t)__builtin_popcountll(marching);
                }
                if (state == ST_MARCH) {
                    bool ended = false, collide = false;
                    float sigma_bar = 0.0f, sigma_low = 0.0f;
                    if (!cell_in_grid(sc, dda)) {
                        ended = true;
                    } else {
                        const uint32_t ci = cell_index(sc, dda);
                        const uint32_t M = lds_maj[ci];
                        if (M != 0u) {
                            const uint32_t q = ((uint32_t)lds_codes[ci >> 2] >> ((ci & 3u) * 2u)) & 3u;
                            sigma_low = sigma_table[(q * M) >> 2].x;
                            const float2 sb = sigma_table[M];
                            sigma_bar = sb.x;
                            const float u = u24_to_float(lcg24(seed));
                            const float dt = -logf_above_one(1.0f - u) * sb.y;
                            const float t_exit = fminf(fminf(dda.tmax.x, dda.tmax.y), dda.tmax.z);
                            if (dda.t + dt < t_exit) {
                                dda.t = dda.t + dt;
                                collide = true;
                            }
                        }
                        if (!collide) {
                            dda_cross(dda, dir);
                        }
                    }
                    if (collide) {
                        const f3 p = mk3(fmaf(dir.x, dda.t, dda.org.x), fmaf(dir.y, dda.t, dda.org.y), fmaf(dir.z, dda.t, dda.org.z));
                        const float z = u24_to_float(lcg24(seed));
                        bool real = z * sigma_bar < sigma_low;
                        if (!real) {
                            uint32_t meta_unused;
                            const uint2 cell = fetch_cell_in_grid(sc, sc.dbricks, p, meta_unused);
                            c_dl += 1;
                            real = z * sigma_bar < filter_at(sc, cell, p) * sc.density_multiplier;
                        }
                        if (real) {
                            if (in_box(sc, p)) {
                                slotA[slot] = make_uint4(__float_as_uint(p.x), __float_as_uint(p.y), __float_as_uint(p.z), seed);
                                collided = true;
                                state = ST_IDLE;
                            } else {
                                ended = true;
                            }
                        }
                    }
                    if (ended) {
                        const uint4 c = slotC[slot];
                        ba.frames[c.w] = make_float4(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), 1.f);
                        if (STATS) {
                            iv_written += 1;
                        }
                        state = ST_IDLE;   // (the slot stays with the lane, empty)
                    }
                }
                if (--burst == 0u) {
                    break;
                }
            }
        }
        // the collided paths wait on BOUNCE; their lanes are free for another path
        {
            const uint64_t cm = __builtin_amdgcn_ballot_w64(collided);
            if (collided) {
                listB[nb + lane_rank(cm)] = (uint8_t)slot;
                slot = kXNoSlot;
            }
            nb += (uint32_t)__builtin_popcountll(cm);
        }

        // ---------------- (4) exit: no job left, no path in a lane or on a list ----------------
        if (drained && q_next == q_end && nb == 0u && nt == 0u && __builtin_amdgcn_ballot_w64(state == ST_MARCH) == 0ull) {
            break;
        }
    }

    uint32_t vals[3] = { c_dl, c_il, c_cap };
#pragma unroll
    for (int i = 0; i < 3; i++) {
        uint32_t v = vals[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v += __shfl_xor(v, off);
        }
        vals[i] = v;
    }
    if (STATS) {
        uint32_t sv[2] = { iv_dealt, iv_written };
#pragma unroll
        for (int i = 0; i < 2; i++) {
            uint32_t v = sv[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                v += __shfl_xor(v, off);
            }
            sv[i] = v;
        }
        if (lane == 0) {
            atomicAdd(&ba.stats[0], (unsigned long long)st_regen);
            atomicAdd(&ba.stats[1], (unsigned long long)st_regen_l);
            atomicAdd(&ba.stats[2], (unsigned long long)st_march);
            atomicAdd(&ba.stats[3], (unsigned long long)st_march_l);
            atomicAdd(&ba.stats[4], (unsigned long long)st_scat);
            atomicAdd(&ba.stats[5], (unsigned long long)st_scat_l);
            atomicAdd(&ba.stats[33], (unsigned long long)st_adopt);
            atomicAdd(&ba.stats[34], (unsigned long long)st_adopt_l);
            atomicAdd(&ba.stats[64], (unsigned long long)sv[0]);
            atomicAdd(&ba.stats[66], (unsigned long long)sv[1]);
        }
    }
    if (lane == 0) {
        atomicAdd(&ba.counters[2], (unsigned long long)vals[0]);
        atomicAdd(&ba.counters[3], (unsigned long long)vals[1]);
        atomicAdd(&ba.counters[4], (unsigned long long)vals[1]);
        atomicAdd(&ba.counters[5], (unsigned long long)vals[2]);
        atomicAdd(&ba.counters[6], (unsigned long long)vals[0]);
        atomicAdd(&ba.counters[7], (unsigned long long)vals[1]);
    }
}

inline size_t delta_w_pool_bytes(uint32_t slots_per_wave)
{
    return (size_t)slots_per_wave * kXWaves * 48u + (size_t)kXWaves * 3u * kWListCap;
}

hipError_t launch_render_delta_w(const DevScene &sc, const BatchArgs &ba, LaunchShape shape, hipStream_t stream)
{
    const dim3 grid(shape.blocks), block(kXThreads);
    const uint32_t spw = shape.pool_slots;
    const size_t dyn = delta_w_pool_bytes(spw);
#define CT_W_LAUNCH(M, S) hipLaunchKernelGGL((render_delta_w_kernel<M, S>), grid, block, dyn, stream, sc, ba, spw)
    if (shape.stats) {
        switch (sc.mode) {
        case 0: CT_W_LAUNCH(0, true); break;
        case 1: CT_W_LAUNCH(1, true); break;
        default: CT_W_LAUNCH(2, true); break;
        }
    } else {
        switch (sc.mode) {
        case 0: CT_W_LAUNCH(0, false); break;
        case 1: CT_W_LAUNCH(1, false); break;
        default: CT_W_LAUNCH(2, false); break;
        }
    }
#undef CT_W_LAUNCH
    return hipGetLastError();
}

// One block per CU; every wave gets an equal share of what the tables leave of the CU's 160 KiB.
LaunchShape wave_exchange_shape(int device)
{
    LaunchShape s{ 256, kXThreads, false };
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
        s.blocks = prop.multiProcessorCount;
    }
    hipFuncAttributes fa{};
    size_t fixed = 80 * 1024;
    if (hipFuncGetAttributes(&fa, (const void *)render_delta_w_kernel<0, false>) == hipSuccess) {
        fixed = fa.sharedSizeBytes;
    }
    const size_t room = 160 * 1024 - fixed - (size_t)kXWaves * 3u * kWListCap;
    uint32_t spw = std::min((uint32_t)(room / (48u * kXWaves)), kWListCap - 1u);
    if (const char *e = getenv("CT_X_SLOTS")) {   // per wave here
        const int v = atoi(e);
        if (v >= 64 && (uint32_t)v <= spw) {
            spw = (uint32_t)v;
        }
    }
    s.pool_slots = spw;
    const int dyn = (int)delta_w_pool_bytes(spw);
    (void)hipFuncSetAttribute((const void *)render_delta_w_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    (void)hipFuncSetAttribute((const void *)render_delta_w_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    (void)hipFuncSetAttribute((const void *)render_delta_w_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    (void)hipFuncSetAttribute((const void *)render_delta_w_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    (void)hipFuncSetAttribute((const void *)render_delta_w_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    (void)hipFuncSetAttribute((const void *)render_delta_w_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    (void)hipGetLastError();
    return s;
}
