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
constexpr unsigned long long kXWatchdogTicks = 3000000000ull;   // 30 s of the 100 MHz wall clock

struct XCtl {
    uint32_t b_tail, b_head;   // BOUNCE ring: collided paths waiting for a scatter batch
    uint32_t t_tail, t_head;   // TRACK ring: scattered paths (and empty slots) waiting for a tracking lane
    int32_t live;              // paths of this block that have started and not ended
    uint32_t drained_waves;    // waves that will not start another sample
    uint32_t pad[2];
};

#define X_FENCE_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#define X_FENCE_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")

CT_DEV uint32_t x_load(const uint32_t *p)
{
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
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
__global__ __launch_bounds__(kXThreads) void render_delta_x_kernel(DevScene sc, BatchArgs ba, uint32_t n_slots)
{
    __shared__ MieLds lds;
    __shared__ uint32_t maj_words[kMajCellsMax / 4];
    __shared__ uint8_t lds_codes[kMajCellsMax / 4];
    __shared__ float2 sigma_table[256];
    __shared__ XCtl ctl;
    extern __shared__ uint4 x_pool[];
    uint4 *const slotA = x_pool, *const slotB = slotA + n_slots, *const slotC = slotB + n_slots;
    uint16_t *const ringB = (uint16_t *)(slotC + n_slots), *const ringT = ringB + kXRing;
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
        // the slots beyond the resident ones start in the TRACK ring as empty slots (lap 0: parity 0); every other ring
        // entry carries the parity of "lap -1"
        const uint32_t spare = n_slots - (uint32_t)kXThreads;
        for (uint32_t i = threadIdx.x; i < kXRing; i += blockDim.x) {
            ringB[i] = 0x8000u;
            ringT[i] = (i < spare) ? (uint16_t)((uint32_t)kXThreads + i) : (uint16_t)0x8000u;
        }
        for (uint32_t i = threadIdx.x; i < n_slots; i += blockDim.x) {
            slotB[i] = make_uint4(0u, 0u, 0u, kXEmpty);
        }
        if (threadIdx.x == 0) {
            ctl.b_tail = ctl.b_head = 0u;
            ctl.t_head = 0u;
            ctl.t_tail = spare;
            ctl.live = 0;
            ctl.drained_waves = 0u;
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

    for (;;) {
        visit += 1;
        if ((visit & 1023u) == 0u && wall_clock64() - t_start > kXWatchdogTicks) {
            bad = true;
        }
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
            break;   // give up (results are wrong): the grid must drain whatever happened
        }
        // ---------------- (1) scatter duty: a batch of collided paths, whichever lanes they collided in ----------------
        {
            const uint32_t pending = x_load(&ctl.b_tail) - x_load(&ctl.b_head);
            const bool marching_any = __builtin_amdgcn_ballot_w64(state == ST_MARCH) != 0ull;
            if (pending >= 64u || (pending != 0u && !marching_any && drained && q_next == q_end)) {
                uint32_t id = 0;
                bool got;
                const uint32_t n = x_pop(ringB, &ctl.b_head, &ctl.b_tail, ~0ull, lane, id, got, bad);
                bool onward = false;
                if (STATS && n != 0u) {
                    st_scat += 1;
                    st_scat_l += n;
                }
                if (got) {
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
                        slotB[id].w = kXEmpty;
                    }
                }
                if (n != 0u) {
                    const uint32_t n_onward = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(onward));
                    ended_paths += (int32_t)(n - n_onward);
                    // a finished path's slot goes on as an empty one while some wave may still start samples
                    const bool all_drained = x_load(&ctl.drained_waves) == (uint32_t)kXWaves;
                    x_push(ringT, &ctl.t_tail, got && (onward || !all_drained), id, lane);
                }
            }
        }

        // ---------------- (2) refill: lanes without a slot take one from the TRACK ring ----------------
        {
            const uint64_t needy = __builtin_amdgcn_ballot_w64(state == ST_IDLE && slot == kXNoSlot);
            if (needy != 0ull && x_load(&ctl.t_tail) != x_load(&ctl.t_head)) {
                uint32_t id = 0;
                bool got;
                const uint32_t n = x_pop(ringT, &ctl.t_head, &ctl.t_tail, needy, lane, id, got, bad);
                if (STATS && n != 0u) {
                    st_adopt += 1;
                    st_adopt_l += n;
                }
                if (got) {
                    slot = id;
                    const uint4 b = slotB[id];
                    if ((b.w & kXEmpty) == 0u) {
                        const uint4 a = slotA[id];
                        dir = mk3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
                        seed = a.w;
                        dda_begin(sc, dda, mk3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)), dir);
                        state = ST_MARCH;
                    }
                }
            }
        }
        // ---------------- (2b) lanes with an empty slot start a new sample (as render_delta_kernel regenerates) ----------------
        {
            const bool empty = state == ST_IDLE && slot != kXNoSlot;
            const uint64_t idle = __builtin_amdgcn_ballot_w64(empty);
            const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
            if (n_idle != 0u && drained && q_next == q_end) {
                // nothing left to start: the slots go out of circulation, the lanes can adopt queued paths
                if (empty) {
                    slot = kXNoSlot;
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
                            const uint32_t out_idx = ba.frame_stride ? ba.out_offset + s * ba.frame_stride + (g * 64u + l) : pixel;
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
                    ended_paths += (int32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ended));
                }
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
        if (ended_paths != 0) {
            if (lane == 0) {
                __hip_atomic_fetch_add(&ctl.live, -ended_paths, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            ended_paths = 0;
        }
        if (drained && q_next == q_end) {
            if (!announced) {
                announced = true;
                if (lane == 0) {
                    __hip_atomic_fetch_add(&ctl.drained_waves, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (__builtin_amdgcn_ballot_w64(state == ST_MARCH) == 0ull) {
                const bool all_drained = x_load(&ctl.drained_waves) == (uint32_t)kXWaves;
                const int32_t live = (int32_t)x_load((const uint32_t *)&ctl.live);
                if (all_drained && live == 0) {
                    break;
                }
                if (x_load(&ctl.b_tail) == x_load(&ctl.b_head) && x_load(&ctl.t_tail) == x_load(&ctl.t_head)) {
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
    return (size_t)n_slots * 48u + 2u * kXRing * sizeof(uint16_t);
}

hipError_t launch_render_delta_x(const DevScene &sc, const BatchArgs &ba, LaunchShape shape, hipStream_t stream)
{
    const dim3 grid(shape.blocks), block(kXThreads);
    const uint32_t n_slots = shape.pool_slots;
    const size_t dyn = delta_x_pool_bytes(n_slots);
#define CT_X_LAUNCH(M, S)                                                                                                   \
    do {                                                                                                                    \
        static bool attr_set = false;                                                                                       \
        if (!attr_set) {                                                                                                    \
            (void)hipFuncSetAttribute((const void *)render_delta_x_kernel<M, S>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)(160 * 1024));                                                                   \
            attr_set = true;                                                                                                \
        }                                                                                                                   \
        hipLaunchKernelGGL((render_delta_x_kernel<M, S>), grid, block, dyn, stream, sc, ba, n_slots);                       \
    } while (0)
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
    const size_t room = 160 * 1024 - fixed - 2u * kXRing * sizeof(uint16_t);
    uint32_t slots = (uint32_t)(room / 48u);
    slots = std::min(slots & ~63u, (uint32_t)kXThreads + kXRing);
    if (const char *e = getenv("CT_X_SLOTS")) {
        const int v = atoi(e);
        if (v >= kXThreads && (uint32_t)v <= slots) {
            slots = (uint32_t)v & ~63u;
        }
    }
    s.pool_slots = slots;
    return s;
}
