// The tile-major LDS-staged frames kernel (gfx950): units of one candidate tile's consecutive time chunks, row tiles shared by
// neighbouring chunks carried over in LDS.  The default for batches too small to give every workgroup two units of one chunk.
// Arithmetic contract and the roles of the waves: mg_frames_common.h.
#include "mg_frames_common.h"

template <int KK, bool LAT_F64, bool FUSE_GMM, bool SPLIT>
__global__ __launch_bounds__(MG_WS_BLOCK) void mg_frames_ws_kernel(
    const float *__restrict__ Epack,      // [RT][KK/2][64][2]
    const float *__restrict__ mean32,     // [RT*16]
    const double *__restrict__ Erpack,    // [RRT][KK][64]
    const double *__restrict__ meanroot,  // [RRT*16]
    const void *__restrict__ lat,         // (B, ld) f32 or f64
    const int32_t *__restrict__ i0tab,    // (T)
    const float4 *__restrict__ w32,       // (T)
    const double *__restrict__ wtap,      // [n_chunks][2][2][64] banded tap weights as f64 MFMA A fragments
    const float4 *__restrict__ rootm,     // (T, 2): SPLIT (the mean/delta split of the root channels): {Mhi, Mlo} per time sample
    const mg_chunk *__restrict__ chunks,
    float *__restrict__ out,              // (B,T,D)
    const double *__restrict__ gPpack,    // FUSE_GMM: precision-Cholesky fragments [K][JT][KK][64]
    const double *__restrict__ gmP,       // FUSE_GMM: mu_k P_k [K][JT*16]
    const double *__restrict__ gcst,      // FUSE_GMM: per-component constants [K]
    float *__restrict__ logp,             // FUSE_GMM: (B) log p(s_b)
    const mg_frames_args a, const int gK, const int gJT, const int buf_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stride = a.stride, D = a.D, Dp = a.Dp, L = a.L, nroot = a.nroot;
    const int root_stride = a.max_wi * nroot + 1;
    const int nbuf = a.nbuf;
    const int max_nt = a.max_nt;
    const int MG_RO_BYTES = MG_RO_BYTES_N(max_nt), MG_TB_BYTES = MG_TB_BYTES_N(max_nt);
    unsigned char *ro_base = smem + nbuf * (size_t)buf_bytes;         // root outputs, one per ring slot
    unsigned char *tb_base = ro_base + nbuf * MG_RO_BYTES;            // per-sample tables, one per ring slot
    unsigned char *rs_base = tb_base + nbuf * MG_TB_BYTES;            // float64 root image (wave 0 only)
    mg_lds_int *prog = (mg_lds_int *)(rs_base + (size_t)MG_NCAND * root_stride * 8);   // 32 counters: progress, GMM
    if (tid < 32) prog[tid] = (tid == 26 || tid == 27) ? 0x7fffffff : 0;   // 26, 27: padding of the gfin wait
    __syncthreads();

    const int64_t U = (int64_t)a.n_tiles * a.n_chunks;
    const int64_t u_begin = (int64_t)blockIdx.x * U / gridDim.x;
    const int64_t u_end = ((int64_t)blockIdx.x + 1) * U / gridDim.x;
    const int n_units = (int)(u_end - u_begin);
    mg_cursor cur;
    cur.tile = (int)(u_begin / a.n_chunks);
    cur.chunk = (int)(u_begin - (int64_t)cur.tile * a.n_chunks);
    // When this workgroup owns whole tiles it walks each tile's chunks starting at chunk (blockIdx mod n_chunks): the
    // workgroups run nearly in lockstep, and without the rotation all 256 of them request the same E' rows from L2 at
    // the same time, the cold first unit above all (-2 % kernel time; MG_DEBUG_FLAGS & 128 switches it off).
    const int rot = (!MG_DBG(128) && cur.chunk == 0 && (n_units % a.n_chunks) == 0) ? (int)(blockIdx.x % a.n_chunks) : 0;
    const int cl = lane & 15, g = lane >> 4;

    if (wave >= MG_WS_NPW) {
        // ================= consumers =================
        const int cj = wave - MG_WS_NPW;                  // candidates cj and cj + 8
        MG_STAMP_DECL
        const int nql = (D - nroot + 3) >> 2;             // quad lanes per sample
        const int gl = nql + 1;                           // + the root lane
        const int rpi = 64 / gl;                          // samples per wave instruction
        // Three samples of 20 lanes: the third one sits in lanes 44 .. 63, not 40 .. 59.  ds_read_b128 serves the lanes in four
        // groups of sixteen ({36-43, 48-51, 60-63} is one), and a row of 80 floats wraps around the 64 banks: the second sample's
        // last quads (columns 68 .. 79, lanes 36-38) and the third sample's first ones (columns 4 .. 15, lanes 40-42) are different
        // addresses in the same banks whenever the two samples share their tap rows -- one extra LDS cycle on most tap reads.
        // Shifted, lanes 60-62 read the addresses lanes 36-38 read (a broadcast).
        const bool shift3 = MG_SWEEP_LANEMAP && gl == 20;
        const int lane_s = (shift3 && lane >= 40) ? lane - 4 : lane;
        const int fsub = lane_s / gl, ql = lane_s - fsub * gl;
        const bool lane_on = shift3 ? (lane < 40 || lane >= 44) : lane < rpi * gl;
        const bool root_lane = ql == nql;
        const int d0 = root_lane ? 0 : nroot + 4 * ql;    // first channel of this lane
        const int nst = root_lane ? nroot : (D - d0 < 4 ? D - d0 : 4);
        const int64_t TD = (int64_t)a.T * D;
        const int dp4 = Dp * 4;
        // byte offset of the lane's quad inside a basis row; SPLIT: the root lane takes the row's first quad, {padding, root channels}
        // (float64 pipeline: the root lane reads what its row's first quad lane reads -- a broadcast -- and so holds channel 3, the
        // fourth float of its store, without asking another lane for it)
        const int lane_img = (SPLIT && root_lane) ? 0 : ((root_lane ? nroot : d0) + a.cshift) * 4;
        const int lane_out = fsub * D + d0;               // float offset inside a row group
        // When every quad lane holds four floats and the root lane three (D = 79: 3 + 19 x 4), the root lane borrows the
        // row's channel 3 from quad lane 0 (a cross-lane read) and ALL lanes store four floats with one instruction;
        // the float written twice carries the same value.  Otherwise lanes store 4 / 3 / 2 / 1 floats by class.
        const bool all4 = nroot == 3 && ((D - nroot) & 3) == 0 && !MG_DBG(8192);
        const int q0_lane = (lane - nql) << 2;            // byte index of this row's quad lane 0 for ds_bpermute
        const unsigned lane_out_b = (unsigned)lane_out * 4u;
        int slot = 0;
        for (int u = 0; u < n_units; u++) {
            MG_STAMP(0);
            const mg_unit un_prev = mg_unit_at(chunks, a, cur, rot);
            mg_cursor_next(cur, a.n_chunks);
            mg_wait_producers(prog, u + 1);
            MG_STAMP(1);
            if (!MG_DBG(2) && cj < un_prev.ncand) {
                const mg_chunk &ck = un_prev.ck;
                const unsigned char *img = smem + (size_t)slot * buf_bytes;
                const float *lds_ro = (const float *)(ro_base + (size_t)slot * MG_RO_BYTES);
                const float *lds_m = lds_ro;   // SPLIT: {Mhi, Mlo} per sample where the root outputs would be
                const float4 *lds_w = (const float4 *)(tb_base + (size_t)slot * MG_TB_BYTES);
                const int *lds_mo = (const int *)(lds_w + max_nt);
                const int col0 = ck.imin * Dp - ck.rt0 * 16;
                const bool has1 = cj + MG_WS_NCW < un_prev.ncand;
                const int c1 = has1 ? cj + MG_WS_NCW : cj;
                const unsigned char *img0 = img + (size_t)(cj * stride + col0) * 4 + lane_img;
                const unsigned char *img1 = img + (size_t)(c1 * stride + col0) * 4 + lane_img;
                const float *ro0 = lds_ro + cj * MG_RO_CS(max_nt), *ro1 = lds_ro + c1 * MG_RO_CS(max_nt);
                float *or0 = out + (size_t)(un_prev.b0 + cj) * TD + (size_t)ck.t0 * D;   // wave-uniform row bases
                float *or1 = out + (size_t)(un_prev.b0 + c1) * TD + (size_t)ck.t0 * D;
                // two row groups x two candidates in flight per trip; the loop exists twice: with the usual row pitch
                // (Dp = 80 floats) as a constant, and with a run-time pitch
                auto sweep_rows = [&](auto pitch_tag, auto all4_tag) {
                constexpr int DP4 = decltype(pitch_tag)::value;
                constexpr bool ALL4 = decltype(all4_tag)::value;   // the usual shape as a constant: every lane stores four floats
                const bool all4l = ALL4 ? true : all4;
                int f_first = 0;
                if constexpr (ALL4 && !SPLIT && MG_SWEEP_FAST) {
                    // the trips whose six samples all lie inside the chunk, lean (see the chunk-stationary kernel): same operations on
                    // the same values as the general loop below
                    if (!MG_DBG(4 | 8192 | 131072 | 2048)) {
                        if (lane_on) {
                            for (; f_first + 2 * rpi <= ck.nT; f_first += 2 * rpi) {
                                const int fa_ = f_first + fsub, fb_ = fa_ + rpi;
                                const float4 wa = lds_w[fa_], wb = lds_w[fb_];
                                const int moa = lds_mo[fa_], mob = lds_mo[fb_];
                                const mg_tap_rows r0a = mg_quad_load<DP4>(img0 + moa, dp4), r0b = mg_quad_load<DP4>(img0 + mob, dp4);
                                const mg_tap_rows r1a = mg_quad_load<DP4>(img1 + moa, dp4), r1b = mg_quad_load<DP4>(img1 + mob, dp4);
                                __builtin_amdgcn_sched_barrier(0);
                                f32x4 v0a = mg_quad_fma(r0a, wa), v0b = mg_quad_fma(r0b, wb);
                                f32x4 v1a = mg_quad_fma(r1a, wa), v1b = mg_quad_fma(r1b, wb);
                                if (root_lane) {
                                    v0a = mg_root_merge(v0a, ro0 + fa_ * 4); v0b = mg_root_merge(v0b, ro0 + fb_ * 4);
                                    v1a = mg_root_merge(v1a, ro1 + fa_ * 4); v1b = mg_root_merge(v1b, ro1 + fb_ * 4);
                                }
                                float *pa0 = or0 + (size_t)f_first * D, *pa1 = or1 + (size_t)f_first * D;          // uniform
                                mg_store4_s(pa0, lane_out_b, v0a);
                                mg_store4_s(pa0 + (size_t)rpi * D, lane_out_b, v0b);
                                if (has1) {
                                    mg_store4_s(pa1, lane_out_b, v1a);
                                    mg_store4_s(pa1 + (size_t)rpi * D, lane_out_b, v1b);
                                }
                            }
                        } else {
                            f_first = ck.nT / (2 * rpi) * (2 * rpi);
                        }
                    }
                }
                for (int f0 = f_first; f0 < ck.nT; f0 += 2 * rpi) {
                    const int fla = f0 + fsub, flb = fla + rpi;
                    const bool oa = lane_on && fla < ck.nT, ob = lane_on && flb < ck.nT;
                    const int fa_ = fla < ck.nT ? fla : ck.nT - 1, fb_ = flb < ck.nT ? flb : ck.nT - 1;
                    float *pa0 = or0 + (size_t)f0 * D, *pa1 = or1 + (size_t)f0 * D;          // uniform
                    float *pb0 = pa0 + (size_t)rpi * D, *pb1 = pa1 + (size_t)rpi * D;
                    if (f0 + rpi < ck.nT || MG_DBG(2048)) {   // the usual trip: both row groups (flag 2048: always)
                        f32x4 v0a, v0b, v1a, v1b;
                        if (MG_DBG(4)) {   // ablation: stores only
                            v0a = v0b = v1a = v1b = f32x4{1.f, 2.f, 3.f, 4.f};
                        } else if (SPLIT || !root_lane) {
                            const float4 wa = lds_w[fa_], wb = lds_w[fb_];
                            const int moa = lds_mo[fa_], mob = lds_mo[fb_];
                            mg_rootm ma, mb;   // SPLIT: every lane asks (a broadcast read), the root lanes use them
                            if constexpr (SPLIT) { ma = mg_rootm_load(lds_m, fa_); mb = mg_rootm_load(lds_m, fb_); }
                            if (MG_DBG(131072)) {
                                v0a = mg_quad_taps_t<DP4>(img0 + moa, wa, dp4);
                                v0b = mg_quad_taps_t<DP4>(img0 + mob, wb, dp4);
                                v1a = mg_quad_taps_t<DP4>(img1 + moa, wa, dp4);
                                v1b = mg_quad_taps_t<DP4>(img1 + mob, wb, dp4);
                            } else {   // all 16 tap rows are requested before the first FMA
                                const mg_tap_rows r0a = mg_quad_load<DP4>(img0 + moa, dp4), r0b = mg_quad_load<DP4>(img0 + mob, dp4);
                                const mg_tap_rows r1a = mg_quad_load<DP4>(img1 + moa, dp4), r1b = mg_quad_load<DP4>(img1 + mob, dp4);
                                __builtin_amdgcn_sched_barrier(0);
                                v0a = mg_quad_fma(r0a, wa);
                                v0b = mg_quad_fma(r0b, wb);
                                v1a = mg_quad_fma(r1a, wa);
                                v1b = mg_quad_fma(r1b, wb);
                            }
                            if constexpr (SPLIT) {
                                if (root_lane) {
                                    if constexpr (ALL4) {   // (all4: three root channels in columns 1 .. 3)
                                        v0a = mg_root_finish<1>(v0a, ma); v0b = mg_root_finish<1>(v0b, mb);
                                        v1a = mg_root_finish<1>(v1a, ma); v1b = mg_root_finish<1>(v1b, mb);
                                    } else {
                                        v0a = mg_root_finish_rt(v0a, ma, a.cshift); v0b = mg_root_finish_rt(v0b, mb, a.cshift);
                                        v1a = mg_root_finish_rt(v1a, ma, a.cshift); v1b = mg_root_finish_rt(v1b, mb, a.cshift);
                                    }
                                }
                            }
                        } else {
                            v0a = *(const f32x4 *)&ro0[fa_ * 4];
                            v0b = *(const f32x4 *)&ro0[fb_ * 4];
                            v1a = *(const f32x4 *)&ro1[fa_ * 4];
                            v1b = *(const f32x4 *)&ro1[fb_ * 4];
                        }
                        if (all4l && MG_DBG(4)) {
                            if (oa) mg_store4_at(pa0, lane_out_b, v0a);
                            if (ob) mg_store4_at(pb0, lane_out_b, v0b);
                            if (oa && has1) mg_store4_at(pa1, lane_out_b, v1a);
                            if (ob && has1) mg_store4_at(pb1, lane_out_b, v1b);
                        } else if (all4l) {
                            const float b0a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v0a[0])));
                            const float b0b = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v0b[0])));
                            const float b1a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v1a[0])));
                            const float b1b = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v1b[0])));
                            if (root_lane) { v0a[3] = b0a; v0b[3] = b0b; v1a[3] = b1a; v1b[3] = b1b; }
                            if (oa) mg_store4_at(pa0, lane_out_b, v0a);
                            if (ob) mg_store4_at(pb0, lane_out_b, v0b);
                            if (oa && has1) mg_store4_at(pa1, lane_out_b, v1a);
                            if (ob && has1) mg_store4_at(pb1, lane_out_b, v1b);
                        } else {
                            if (oa) mg_store_n(pa0 + lane_out, v0a, nst);
                            if (ob) mg_store_n(pb0 + lane_out, v0b, nst);
                            if (oa && has1) mg_store_n(pa1 + lane_out, v1a, nst);
                            if (ob && has1) mg_store_n(pb1 + lane_out, v1b, nst);
                        }
                    } else {                                     // the chunk's last rows fill one group only: half the work
                        f32x4 v0a, v1a;
                        if (SPLIT || !root_lane) {
                            const float4 wa = lds_w[fa_];
                            const int moa = lds_mo[fa_];
                            mg_rootm ma;
                            if constexpr (SPLIT) ma = mg_rootm_load(lds_m, fa_);
                            v0a = mg_quad_taps_t<DP4>(img0 + moa, wa, dp4);
                            v1a = mg_quad_taps_t<DP4>(img1 + moa, wa, dp4);
                            if constexpr (SPLIT) {
                                if (root_lane) {
                                    if constexpr (ALL4) { v0a = mg_root_finish<1>(v0a, ma); v1a = mg_root_finish<1>(v1a, ma); }
                                    else { v0a = mg_root_finish_rt(v0a, ma, a.cshift); v1a = mg_root_finish_rt(v1a, ma, a.cshift); }
                                }
                            }
                        } else {
                            v0a = *(const f32x4 *)&ro0[fa_ * 4];
                            v1a = *(const f32x4 *)&ro1[fa_ * 4];
                        }
                        if (all4l) {
                            const float b0a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v0a[0])));
                            const float b1a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(q0_lane, __builtin_bit_cast(int, v1a[0])));
                            if (root_lane) { v0a[3] = b0a; v1a[3] = b1a; }
                            if (oa) mg_store4_at(pa0, lane_out_b, v0a);
                            if (oa && has1) mg_store4_at(pa1, lane_out_b, v1a);
                        } else {
                            if (oa) mg_store_n(pa0 + lane_out, v0a, nst);
                            if (oa && has1) mg_store_n(pa1 + lane_out, v1a, nst);
                        }
                    }
                }
                };
                if (dp4 == 320 && all4 && !MG_DBG(16384)) sweep_rows(std::integral_constant<int, 320>{}, std::true_type{});
                else sweep_rows(std::integral_constant<int, 0>{}, std::false_type{});
            }
            MG_STAMP(4);
            mg_publish(prog, wave, lane, u + 1);
            if (++slot == nbuf) slot = 0;
            MG_STAMP(5);
        }
        MG_STAMP_DUMP;
    } else if (wave != 0) {
        // ================= f32 producers (waves 1..3): E' fragments -> MFMA -> LDS image =================
        float sfrag[KK];
#pragma unroll
        for (int kk = 0; kk < KK; kk++) sfrag[kk] = 0.f;
        int cur_tile = -1;
        int prev_tile = -1, prev_chunk = -1;
        const float2 *ep = (const float2 *)Epack;
        int slot = 0;
        MG_STAMP_DECL
        for (int u = 0; u < n_units; u++) {
            MG_STAMP(0);
            const mg_unit un = mg_unit_at(chunks, a, cur, rot);
            mg_cursor_next(cur, a.n_chunks);
            if (u >= nbuf) mg_wait_consumers(prog, u - nbuf + 1);   // the slot's previous unit has been swept
            MG_STAMP(5);
            if (!MG_DBG(1)) {
                const mg_chunk &ck = un.ck;
                float *lds_c = (float *)(smem + (size_t)slot * buf_bytes);
                if (un.tile != cur_tile) {
                    cur_tile = un.tile;
                    int g_op = g;   // opaque: keeps the (loop-invariant) clamped indices from being hoisted and spilled
                    asm volatile("" : "+v"(g_op));
                    mg_load_sfrag<KK, LAT_F64>(sfrag, lat, un, a.ld, L, cl, g_op);
                }
                // Consecutive chunks of a tile share basis functions (for 'walk' windows of 10 advance by 7): the row tiles this
                // unit has in common with the previous one are copied from the previous slot (LDS -> LDS, the same
                // bits) instead of being recomputed: 37 % fewer MFMAs and E' fragment loads, which is what slows
                // the sweep waves down (matrix-pipe time on the shared SIMDs, L2 requests in the store path).
                int n_ov = 0, src_shift = 0;
                if (un.tile == prev_tile && un.chunk == prev_chunk + 1) {   // the previous unit was this tile's previous chunk
                    const mg_chunk pk = chunks[un.chunk - 1];
                    src_shift = ck.rt0 - pk.rt0;
                    n_ov = pk.rt0 + pk.ntiles - ck.rt0;   // tiles [ck.rt0, pk.rt0 + pk.ntiles) exist in the previous slot
                    n_ov = (n_ov < 0 || src_shift < 0) ? 0 : (n_ov > ck.ntiles ? ck.ntiles : n_ov);   // a grid may run backwards
                }
                if (MG_DBG(4096)) n_ov = 0;   // ablation: no carried-over tiles (every window computed in full)
                // this slot held unit u - nbuf and was the copy source of unit u - nbuf + 1: every row producer must
                // have finished that unit before the slot is overwritten (with two slots: a full meeting per unit)
                if (u >= nbuf - 1) mg_wait_row_producers(prog, u - nbuf + 2);
                mg_produce_f32<KK>(ep, mean32, ck, lds_c, stride, n_ov, wave - 1, MG_WS_NPW - 1, sfrag, lane, cl, g, MG_DBG(256) ? 0 : (int)(blockIdx.x / a.n_chunks),
                                   (MG_DBG(32768) && (u & 1)) ? (a.debug | 512) : a.debug);   // 32768: E' loads on every other unit only
                MG_STAMP(2);
                if (n_ov > 0) {
                    mg_wait_row_producers(prog, u);   // the previous unit's window is complete
                    const float *lds_p = (const float *)(smem + (size_t)(slot == 0 ? nbuf - 1 : slot - 1) * buf_bytes);
                    for (int t = wave - 1; t < n_ov; t += MG_WS_NPW - 1)
                        *(f32x4 *)&lds_c[cl * stride + t * 16 + 4 * g] = *(const f32x4 *)&lds_p[cl * stride + (t + src_shift) * 16 + 4 * g];
                }
                MG_STAMP(3);
            }
            prev_tile = un.tile;
            prev_chunk = un.chunk;
            mg_publish(prog, wave, lane, u + 1);
            if (++slot == nbuf) slot = 0;
            MG_STAMP(4);
        }
        MG_STAMP_DUMP;
    } else {
        // ================= wave 0: tables, root rows and root taps (f64 MFMA), one unit ahead =================
        // per-lane constants of the tap MFMA: B operand = rows[m = 4 ks + (l >> 4)][col = 16 ct + (l & 15)] with
        // col = candidate * nroot + channel (48 columns = 3 tiles); D column = the same col
        int tap_b_off[3][MG_TAP_KS], tap_o_off[3];
        bool tap_b_ok[3][MG_TAP_KS];
#pragma unroll
        for (int ct = 0; ct < 3; ct++) {
            const int col = ct * 16 + cl;
            const bool colok = col < MG_NCAND * nroot;
            const int cc = colok ? col / nroot : 0, cd = colok ? col - cc * nroot : 0;
            tap_o_off[ct] = colok ? cc * MG_RO_CS(max_nt) + cd : -1;
#pragma unroll
            for (int ks = 0; ks < MG_TAP_KS; ks++) {
                const int m = 4 * ks + g;
                tap_b_ok[ct][ks] = colok && m < a.max_wi;
                tap_b_off[ct][ks] = cc * root_stride + m * nroot + cd;
            }
        }
        MG_STAMP_DECL
        auto root_stage = [&](const mg_unit &un, int slot) {   // tables -> tb[slot], root rows -> rs, root outputs -> ro[slot]
            const mg_chunk &ck = un.ck;
            float4 *tw = (float4 *)(tb_base + (size_t)slot * MG_TB_BYTES);
            int *tmo = (int *)(tw + max_nt);
            double *rs = (double *)rs_base;
            float4 r_w = {0.f, 0.f, 0.f, 0.f};
            int r_i0 = 0;
            if (lane < ck.nT) { r_w = w32[ck.t0 + lane]; r_i0 = i0tab[ck.t0 + lane]; }
            if constexpr (SPLIT) {   // the mean/delta split: the unit's tables only; the root lanes of the sweep finish the root channels
                float4 *tm = (float4 *)(ro_base + (size_t)slot * MG_RO_BYTES);
                if (lane < ck.nT) {
                    const float4 mh = rootm[2 * (ck.t0 + lane)], ml = rootm[2 * (ck.t0 + lane) + 1];
                    tw[lane] = r_w;
                    tmo[lane] = (r_i0 - ck.imin) * Dp * 4;
                    tm[2 * lane] = mh;
                    tm[2 * lane + 1] = ml;
                }
                return;
            }
            double r_wt[MG_TAP_FT * MG_TAP_KS];
#pragma unroll
            for (int e = 0; e < MG_TAP_FT * MG_TAP_KS; e++) r_wt[e] = wtap[((size_t)un.chunk * (MG_TAP_FT * MG_TAP_KS) + e) * 64 + lane];
            typename mg_gmm_xt<LAT_F64>::type s64frag[KK];   // widened to float64 at the MFMA
            mg_gmm_load_x<KK, LAT_F64>(s64frag, lat, un.b0, un.ncand, a.ld, L, cl, g);
            if (MG_DBG(32)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); MG_STAMP(1); }
            // up to 3 root tiles (8 basis functions x 3 channels = 24 rows, rr = i*nroot + d), chains
            // interleaved; v_mfma_f64_16x16x4_f64 C/D: col = lane & 15, row = (lane >> 4) + 4*reg
            f64x4 racc[3];
            const double *rpp[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int tc = t < ck.nrt ? t : ck.nrt - 1;
                rpp[t] = Erpack + ((size_t)(ck.rrt0 + tc) * KK) * 64 + lane;
                const int row0 = (ck.rrt0 + tc) * 16;
                racc[t][0] = meanroot[row0 + g];
                racc[t][1] = meanroot[row0 + g + 4];
                racc[t][2] = meanroot[row0 + g + 8];
                racc[t][3] = meanroot[row0 + g + 12];
            }
            // all fragments of the 3 tiles in one round of loads (one L2 round trip under store pressure costs
            // thousands of cycles); only for many components in two halves to stay inside the register budget
            constexpr int NH = KK <= 10 ? 1 : 2;
            constexpr int KH = KK / NH;
#pragma unroll
            for (int h = 0; h < NH; h++) {
                double rp[3][KH];
#pragma unroll
                for (int t = 0; t < 3; t++)
#pragma unroll
                    for (int q = 0; q < KH; q++) rp[t][q] = rpp[t][(h * KH + q) * 64];
                if (h == 0) {
                    if (lane < ck.nT) {
                        tw[lane] = r_w;
                        tmo[lane] = (r_i0 - ck.imin) * Dp * 4;   // byte offset of the first tap row in the f32 image
                    }
                }
#pragma unroll
                for (int q = 0; q < KH; q++)
#pragma unroll
                    for (int t = 0; t < 3; t++)
                        racc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(rp[t][q], (double)s64frag[h * KH + q], racc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int lr0 = (ck.rrt0 + t) * 16 + g - ck.imin * nroot;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int lr = lr0 + 4 * r;
                    if (t < ck.nrt && lr >= 0 && lr < ck.wi * nroot) rs[cl * root_stride + lr] = racc[t][r];
                }
            }
            if (MG_DBG(32)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); MG_STAMP(2); }
            // root taps, again on the float64 matrix pipe: out[f][(c,d)] = sum_m W[f][m] * rows[m][(c,d)] with the
            // banded W[f][m] = w[f][m - m0(f)] (0 outside the 4 taps) pre-packed per chunk as A fragments.  The zero
            // products leave the accumulator untouched and the taps are met in ascending m, so the result is
            // bit-identical to w0*c0, fma(w1,c1,.), fma(w2,c2,.), fma(w3,c3,.).  Same wave wrote rs: program order syncs.
            float *ro = (float *)(ro_base + (size_t)slot * MG_RO_BYTES);
#pragma unroll
            for (int ft = 0; ft < MG_TAP_FT; ft++) {
                if (ft * 16 < ck.nT) {
                    f64x4 acc[3];
                    double bv[3][MG_TAP_KS];
#pragma unroll
                    for (int ct = 0; ct < 3; ct++) {
                        acc[ct] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int ks = 0; ks < MG_TAP_KS; ks++)   // rows at or beyond this chunk's window were never written: 0 * stale LDS could be NaN
                            bv[ct][ks] = (tap_b_ok[ct][ks] && 4 * ks + g < ck.wi) ? rs[tap_b_off[ct][ks]] : 0.0;
                    }
#pragma unroll
                    for (int ks = 0; ks < MG_TAP_KS; ks++)
#pragma unroll
                        for (int ct = 0; ct < 3; ct++)
                            acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(r_wt[ft * MG_TAP_KS + ks], bv[ct][ks], acc[ct], 0, 0, 0);
                    // D[row = f = 16 ft + (l >> 4) + 4 reg][col]
#pragma unroll
                    for (int ct = 0; ct < 3; ct++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int fo = ft * 16 + g + 4 * r;
                            if (tap_o_off[ct] >= 0 && fo < ck.nT) ro[tap_o_off[ct] + fo * 4] = (float)acc[ct][r];
                        }
                }
                if (MG_DBG(32) && ft == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); MG_STAMP(6); }
            }
        };
        int slot = 0;
        for (int u = 0; u < n_units; u++) {
            MG_STAMP(0);
            const mg_unit un = mg_unit_at(chunks, a, cur, rot);
            mg_cursor_next(cur, a.n_chunks);
            if (u >= nbuf) mg_wait_consumers(prog, u - nbuf + 1);
            MG_STAMP(5);
            if (!MG_DBG(1) && !MG_DBG(1024)) root_stage(un, slot);
            MG_STAMP(3);
            mg_publish(prog, wave, lane, u + 1);
            if (++slot == nbuf) slot = 0;
            MG_STAMP(4);
        }
        MG_STAMP_DUMP;
    }
    if (FUSE_GMM && wave < MG_WS_NPW) {
        // up to four tiles per workgroup, as two groups of two written out one after the other (a loop would let the
        // compiler hoist the exp/log polynomial constants above the MFMA code and spill)
        mg_fused_gmm_terms<KK, LAT_F64>(prog, gPpack, gmP, gcst, lat, a.B, a.ld, L, a.n_tiles, gK, gJT, wave, lane, 0);
        if (wave < 2) mg_fused_gmm_finish(prog, logp, a.B, a.n_tiles, gK, wave, lane, 0);
        const int64_t my_tiles = ((int64_t)blockIdx.x + 1) * a.n_tiles / gridDim.x - (int64_t)blockIdx.x * a.n_tiles / gridDim.x;
        if (my_tiles > 2) {
            mg_wait_producers(prog + 24, 1);   // gfin[0], gfin[1] (entries 2, 3 are preset): both term buffers are free again
            mg_fused_gmm_terms<KK, LAT_F64>(prog, gPpack, gmP, gcst, lat, a.B, a.ld, L, a.n_tiles, gK, gJT, wave, lane, 1);
            if (wave < 2) mg_fused_gmm_finish(prog, logp, a.B, a.n_tiles, gK, wave, lane, 1);
        }
    }
}



// -----------------------------------------------------------------------------------------
// launch
// -----------------------------------------------------------------------------------------
template <int KK, bool LAT_F64, bool FUSE, bool SPLIT>
static int mg_launch_ws_inst(mg_primitive *p, const mg_time_grid *g, const void *lat, float *out, float *logp, const mg_frames_args &a,
                                 int buf_bytes, int lds, int grid, const mg_launch_events &ev) {
    // hipExtLaunchKernelGGL with NULL events is hipLaunchKernelGGL; with events the dispatch records its own begin and end
    hipExtLaunchKernelGGL((mg_frames_ws_kernel<KK, LAT_F64, FUSE, SPLIT>), dim3(grid), dim3(MG_WS_BLOCK), lds, p->ctx->stream, ev.start, ev.stop, 0,
                          (const float *)p->d_Epack, (const float *)p->d_mean32, (const double *)p->d_Erpack, (const double *)p->d_meanroot, lat,
                          (const int32_t *)g->d_i0, (const float4 *)g->d_w32, (const double *)g->d_wtap, (const float4 *)g->d_rootm,
                          (const mg_chunk *)g->d_chunks, out,
                          (const double *)p->d_gPpack, (const double *)p->d_gmPpad, (const double *)p->d_gconst, logp, a, (int)p->K,
                          (int)((p->L + 15) / 16), buf_bytes);
    MG_HIP_CHECK(hipGetLastError());
    return MG_OK;
}

template <int KK>
static int mg_launch_ws_kk(mg_primitive *p, const mg_time_grid *g, const void *lat, float *out, float *logp, const mg_frames_args &a,
                               bool lat_f64, bool split, int buf_bytes, int lds, int grid, const mg_launch_events &ev) {
    if (logp) {
        // fused instances exist for <= 40 components: beyond that the mixture fragments no longer fit the
        // register budget next to the sweep (mg_frames_can_fuse_gmm refuses, so this is never reached)
        if constexpr (KK <= MG_FUSE_MAX_KK) {
            if (split)
                return lat_f64 ? mg_launch_ws_inst<KK, true, true, true>(p, g, lat, out, logp, a, buf_bytes, lds, grid, ev)
                               : mg_launch_ws_inst<KK, false, true, true>(p, g, lat, out, logp, a, buf_bytes, lds, grid, ev);
            return lat_f64 ? mg_launch_ws_inst<KK, true, true, false>(p, g, lat, out, logp, a, buf_bytes, lds, grid, ev)
                           : mg_launch_ws_inst<KK, false, true, false>(p, g, lat, out, logp, a, buf_bytes, lds, grid, ev);
        }
        mg_set_error("mg_step_frames_and_logp: no fused kernel for %d components", p->L);
        return MG_ERR_UNSUPPORTED;
    }
    if (split)
        return lat_f64 ? mg_launch_ws_inst<KK, true, false, true>(p, g, lat, out, nullptr, a, buf_bytes, lds, grid, ev)
                       : mg_launch_ws_inst<KK, false, false, true>(p, g, lat, out, nullptr, a, buf_bytes, lds, grid, ev);
    return lat_f64 ? mg_launch_ws_inst<KK, true, false, false>(p, g, lat, out, nullptr, a, buf_bytes, lds, grid, ev)
                   : mg_launch_ws_inst<KK, false, false, false>(p, g, lat, out, nullptr, a, buf_bytes, lds, grid, ev);
}

int mg_launch_frames_ws(mg_primitive *p, const mg_time_grid *g, const void *lat, float *out, float *logp, const mg_frames_args &a, bool lat_f64,
                        bool split, int buf_bytes, int lds, int grid, const mg_launch_events &ev) {
    switch (p->KK) {
#ifndef MG_ONLY_KK10
        case 2: return mg_launch_ws_kk<2>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
        case 4: return mg_launch_ws_kk<4>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
        case 6: return mg_launch_ws_kk<6>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
        case 8: return mg_launch_ws_kk<8>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
#endif
        case 10: return mg_launch_ws_kk<10>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
#ifndef MG_ONLY_KK10
        case 12: return mg_launch_ws_kk<12>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
        case 14: return mg_launch_ws_kk<14>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
        case 16: return mg_launch_ws_kk<16>(p, g, lat, out, logp, a, lat_f64, split, buf_bytes, lds, grid, ev);
#endif
        default: mg_set_error("mg_back_project_frames: MFMA path needs n_components <= 64"); return MG_ERR_UNSUPPORTED;
    }
}

template <int KK>
static int mg_ws_attr_kk() {
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_ws_kernel<KK, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_ws_kernel<KK, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_ws_kernel<KK, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_ws_kernel<KK, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if constexpr (KK <= MG_FUSE_MAX_KK) {
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_ws_kernel<KK, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_ws_kernel<KK, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_ws_kernel<KK, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MG_HIP_CHECK(hipFuncSetAttribute((const void *)mg_frames_ws_kernel<KK, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    return MG_OK;
}
int mg_frames_ws_attributes() {
    int rc;
#ifndef MG_ONLY_KK10
    if ((rc = mg_ws_attr_kk<2>()) != MG_OK) return rc;
    if ((rc = mg_ws_attr_kk<4>()) != MG_OK) return rc;
    if ((rc = mg_ws_attr_kk<6>()) != MG_OK) return rc;
    if ((rc = mg_ws_attr_kk<8>()) != MG_OK) return rc;
#endif
    if ((rc = mg_ws_attr_kk<10>()) != MG_OK) return rc;
#ifndef MG_ONLY_KK10
    if ((rc = mg_ws_attr_kk<12>()) != MG_OK) return rc;
    if ((rc = mg_ws_attr_kk<14>()) != MG_OK) return rc;
    if ((rc = mg_ws_attr_kk<16>()) != MG_OK) return rc;
#endif
    return MG_OK;
}
