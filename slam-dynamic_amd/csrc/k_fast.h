// k_fast_cells_staged — FAST-9/16 + NMS + threshold fallback per cell, staged so that the expensive
// work only runs on dense survivor lists (src/ORBextractor.cc:789-829, cv::FAST).
//
//   phase 0  window -> LDS tile (unaligned dword loads, 4 B/lane), tile column c = window x + 1 so that
//            the scanned pixels start on a word boundary
//   phase 1  ALL scanned pixels, 4 per thread from packed words: compass test.  Any 9-arc of the
//            16-ring holds two ADJACENT compass points (ring 0,4,8,12), so a corner needs an adjacent
//            compass pair both darker than v-T or both brighter than v+T; done on halved pixels, 4 px per 32-bit op.
//   phase 2  survivors (wave-ballot compacted): cornerScore (9-arc min / max as three 3-arcs) -> score tile; score >= T
//            IS the FAST-9 test at T, so the corners fall out of the score (compacted again)
//   phase 3  3x3 strict-maximum NMS on the corner list -> row bit masks
//   phases 1-3 run at T = iniTh and, only for a cell without a survivor, again at T = minTh (the reference's retry)
//   phase 4  row-major ordered emission (wave prefix sum over the row masks + popcount inside the row)
// Same results as k_fast_cells (kept as the generic path for windows > 52 px); see that kernel's header
// comment for why the threshold-free score map reproduces OpenCV's two-threshold behaviour.
#pragma once
#include "k_extract.h"

#define SD_FS_MAXWIN 52            // largest window side handled here
#define SD_FS_TW 64                // tile row stride (bytes) = four 16-byte chunks (window width + 1 <= 53)
#define SD_FS_SW 48                // score tile row stride (bytes), 1-px zero frame included
#define SD_FS_MAXSCAN (46 * 46)

typedef uint32_t __attribute__((aligned(1))) sd_u32_unaligned;

__device__ __forceinline__ int sd_wave_append(bool pred, int* counter)
{
    // returns the slot for lanes with pred, using one LDS atomic per wave
    const unsigned long long m = __ballot(pred);
    if (m == 0) return -1;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1, 64);
    return pred ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// Slot reservation by ONE LDS atomic per lane (returns the old counter value).  Written as the instruction itself: handed an atomicAdd on a
// workgroup-uniform address the compiler's atomic optimiser rebuilds the wave-level form -- a scan over the active lanes (v_readlane / v_writelane
// per lane, or six DPP adds) around one atomic -- i.e. exactly the vector-ALU work this kernel, bound by vector-ALU issue, wants to hand to the LDS
// unit.  The same-address atomics of a wave serialise there; the LDS unit has the time (its other traffic here: ~20 byte reads per candidate).
__device__ __forceinline__ int sd_lds_add_rtn(int* counter, int v)
{
    int old;
    const unsigned addr = (unsigned)(size_t)counter;          // generic address of an LDS object: the low half is the LDS offset
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(v) : "memory");
    return old;
}

// How the three lists are appended to -- measured on one box, k_fast_cells_staged per 512 images (tools/fast_append_ab.sh, two rounds each):
//   round 3: four bit-plane ballots + mbcnt in phase 1, sd_wave_append (ballot, ffs, shuffle) in phases 2 / 3                      1.50 ms
//   one LDS atomic per LANE everywhere (sd_lds_add_rtn)                                                                            1.33 / 1.38
//   phase 1 one atomic per 16-lane ROW (DPP row scan + bpermute), phases 2 / 3 per lane                                            1.31 / 1.34
//   phase 1 per row, phases 2 / 3 atomicAdd(p, 1) = the compiler's wave-aggregated form (ballot + mbcnt around one atomic)         1.28 / 1.24
//   phase 1 per LANE, phases 2 / 3 atomicAdd(p, 1)                                                                     <- kept     1.25 / 1.25
// Phase 1's count differs per lane (0 .. 4 candidates), where the wave-level forms need a scan; phases 2 / 3 add 1, where the compiler's form is a
// ballot and two mbcnt -- and ~28 lanes of a wave hitting one LDS address cost more than that.
#ifndef SD_FAST_APPEND
#define SD_FAST_APPEND 0                       // 0: one atomic per lane in phase 1; 1: one per 16-lane row
#endif
#ifndef SD_FAST_ADD1
#define SD_FAST_ADD1(p) atomicAdd((p), 1)      // or sd_lds_add_rtn((p), 1): one atomic per lane
#endif

// Per-cell descriptor of the staged kernel: everything the window fetch needs in ONE scalar load (the generic SdCell needs the
// level table behind it, a second dependent load before the first pixel can be requested).
struct SdFastCell {
    uint32_t srcOff;        // byte offset of tile column 0 / window row 0 inside one image's pyramid block
    int stride;             // level row stride
    short ww, wh;           // window size
    short jw, ih;           // ORBextractor.cc:822-823 shift
    int listOffset, cap;
};
struct SdFastArgs {
    unsigned long long pyrImageBytes;
    int cellTotal, nImages, listCap, minTh, iniTh, cellListCap;
};
#define SD_FS_ROWS 48              // row-mask slots (scan height <= 46)

// One workgroup of NT threads per cell.  The cost of a cell is VALU work plus the latency of its window fetch, so small
// workgroups with a small LDS footprint (lists sized by the plan's largest scan area) keep many independent cells in flight
// per CU.
// (Measured: giving a workgroup 2 or 4 consecutive cells — same barriers, fuller lanes in the short phases — is SLOWER:
// 1.02 ms per 256 images here vs 1.48 ms (128 threads x 2 cells), 1.15 ms (256 x 2), 1.61 ms (256 x 4): the LDS footprint
// halves the cells in flight per CU and the phases do not shrink.)
template <int NT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8))) k_fast_cells_staged(const uint8_t* __restrict__ pyr,
                                                          const SdFastCell* __restrict__ cells,
                                                          uint32_t* __restrict__ cellList, int* __restrict__ cellCount,
                                                          const SdFastArgs A)
{
    __shared__ __align__(16) uint8_t tile[SD_FS_MAXWIN * SD_FS_TW];
    __shared__ __align__(16) uint8_t score[SD_FS_SW * SD_FS_SW];
    __shared__ unsigned long long rowAll[SD_FS_ROWS];                       // NMS survivors per scan row, one bit per column
    extern __shared__ __align__(16) unsigned char dyn_smem[];
    unsigned short* list1 = (unsigned short*)dyn_smem;              // [listCap] pixels that pass the compass test
    unsigned short* list2 = list1 + A.listCap;                      // [listCap] corners at minTh
    unsigned* kept = (unsigned*)(list2 + A.listCap);                // [listCap / 4 + 4] NMS maxima: (sy<<16)|(sx<<8)|score
    __shared__ int s_cnt1, s_cnt2, s_cnt3;
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t* tileW = (uint32_t*)tile;
    uint32_t* scoreW = (uint32_t*)score;
    // XCD-aware order: grid = (8 * cells, image groups); the linear workgroup id % 8 = blockIdx.x % 8 picks the XCD
    // (round-robin dispatch), so XCD x walks the cells of images x, x + 8, ... in cell order and neighbouring cells (which
    // share 128-byte lines and a 6-px halo) meet in ONE L2.  Measured at 256 images: FETCH_SIZE per launch 928 -> 173 MiB
    // (x2 = the algorithmic bytes).  (Splitting the cells of ONE image over the XCDs cut the fetch as much but was 35 % slower.)
    const int ci = blockIdx.x >> 3;
    const int img = blockIdx.y * 8 + (blockIdx.x & 7);
    if (img >= A.nImages) return;
    const SdFastCell c = cells[ci];
    const int ww = c.ww, wh = c.wh;
    const int sw = ww - 6, sh = wh - 6;
    if (sw <= 0 || sh <= 0) { if (tid == 0) cellCount[(size_t)img * A.cellTotal + ci] = 0; return; }
    {
        // ---- phase 0: window -> LDS.  All row requests of a thread are issued before the first one is consumed (one memory
        // round trip per cell instead of one per 8 rows); the LDS clears run underneath.
        // 16-byte requests (the window rows are not aligned; the hardware takes unaligned dwordx4): four chunks per row, two requests per
        // thread cover the 52-row maximum -- a quarter of the address arithmetic and LDS writes of the dword form (the kernel is VALU-bound)
        const uint8_t* src = pyr + (size_t)img * A.pyrImageBytes + c.srcOff;
        const int wd = tid & 3, y0 = tid >> 2;
        constexpr int NR = (SD_FS_MAXWIN + NT / 4 - 1) / (NT / 4);
        sd_u4v px[NR];
        const int nq = (ww + 1 + 15) >> 4;                  // chunks that hold window bytes (tile column 0 = window x - 1)
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const int y = y0 + k * (NT / 4);
            if (wd < nq && y < wh) px[k] = *(const sd_u128_unaligned*)(src + (unsigned)__mul24(y, c.stride) + 16 * wd);
        }
        // only the frame and the scanned rows of the score tile are ever read: rows 0 .. sh+1
        for (int i = tid; i < (sh + 2) * (SD_FS_SW / 4); i += NT) scoreW[i] = 0;
        if (tid < SD_FS_ROWS) rowAll[tid] = 0;
        if (tid == 0) { s_cnt1 = 0; s_cnt2 = 0; s_cnt3 = 0; }
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const int y = y0 + k * (NT / 4);
            if (wd < nq && y < wh) *(sd_u4v*)(tile + y * SD_FS_TW + 16 * wd) = px[k];
        }
        __syncthreads();
        // ---- phase 1: compass quick test, 4 px per thread, four pixels per 32-bit operation on HALVED pixel values (7 bits per
        // byte, bit 7 of every byte is the borrow guard of a byte-wise subtraction):  X < v - T  implies
        // (X >> 1) <= (v >> 1) - k  and  X > v + T  implies  (X >> 1) >= (v >> 1) + k  with k = (T + 1) / 2, so the halved test never
        // loses a corner; the few extra survivors (+6 % measured) are rejected by the exact score of phase 2.
        // The adjacent-pair test over the cycle S-E-N-W collapses to (dS|dN) & (dE|dW); same for the brighter side.
        const uint32_t M7 = 0x7f7f7f7fu, G7 = 0x80808080u;
        const int ng = (sw + 3) >> 2;
        const int shift = ng <= 8 ? 3 : 4;                 // items per scan row = 1 << shift (no division)
        const int nitems = sh << shift;
        // The reference's order (ORBextractor.cc:809-816): FAST at iniTh; only a cell that yields NO keypoint is redone at minTh.
        // Most cells of a textured image stop after the first pass, whose survivor lists are ~40 % shorter than minTh's.
        int n3 = 0;
#pragma nounroll
        for (int pass = 0; pass < 2; pass++) {
        const int T = pass ? A.minTh : A.iniTh;
        const uint32_t KG = G7 - (uint32_t)((T + 1) >> 1) * 0x01010101u;
        for (int it0 = 0; it0 < nitems; it0 += NT) {
            const int it = it0 + tid;
            const int sy = it >> shift, gq = it & ((1 << shift) - 1);
            const bool live = it < nitems && gq < ng;
            unsigned pmask = 0;
            if (live) {
                const uint32_t* row = tileW + (sy + 3) * (SD_FS_TW / 4) + gq;
                const uint32_t w0 = row[0], w1 = row[1], w2 = row[2];
                const uint32_t N = tileW[sy * (SD_FS_TW / 4) + gq + 1];
                const uint32_t S = tileW[(sy + 6) * (SD_FS_TW / 4) + gq + 1];
                // halved pixels, 7 bits per byte (the alignbit forms shift the E / W neighbours into place on the way)
                const uint32_t C7 = (w1 >> 1) & M7, N7 = (N >> 1) & M7, S7 = (S >> 1) & M7;
                const uint32_t E7 = __builtin_amdgcn_alignbit(w2, w1, 25) & M7;
                const uint32_t W7 = __builtin_amdgcn_alignbit(w1, w0, 9) & M7;
                const uint32_t loD = C7 + KG;                       // byte = 128 + (v7 - k): bit 7 = "v7 >= k", low bits = v7 - k
                const uint32_t gD = loD | G7;
                const uint32_t loB = (C7 ^ M7) + KG;                // the same in the complemented (127 - x) domain
                const uint32_t hB = (loB | G7) - M7;
                const uint32_t dV = (gD - S7) | (gD - N7), dH = (gD - E7) | (gD - W7);   // bit 7: X7 <= v7 - k
                const uint32_t bV = (S7 + hB) | (N7 + hB), bH = (E7 + hB) | (W7 + hB);   // bit 7: X7 >= v7 + k
                pmask = ((dV & dH & loD) | (bV & bH & loB)) & G7;
                const int nvalid = sw - 4 * gq;                               // pixels of this group inside the scan
                if (nvalid < 4) pmask &= (1u << (8 * nvalid)) - 1u;
            }
            // Slot reservation: ONE LDS atomic per lane that holds candidates (sd_lds_add_rtn; list order is free).  The wave-level form (four
            // bit-plane ballots + mbcnt + the per-plane offsets) was ~45 vector instructions per iteration, more than the compass test itself.
#if SD_FAST_APPEND == 1
            // experiment: candidates counted per 16-lane row on the DPP network, one atomic per row (four per wave) instead of one per lane
            const int cnt = __popc(pmask);
            int inc = cnt;
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);
            int rowBase = 0;
            if ((lane & 15) == 15 && inc) rowBase = sd_lds_add_rtn(&s_cnt1, inc);
            rowBase = __shfl(rowBase, lane | 15, 64);
            if (pmask) {
                int pos = rowBase + inc - cnt;
                const unsigned short e0 = (unsigned short)((sy << 8) | (4 * gq));
#else
            if (pmask) {
                int pos = sd_lds_add_rtn(&s_cnt1, __popc(pmask));
                const unsigned short e0 = (unsigned short)((sy << 8) | (4 * gq));
#endif
                if (pmask & 0x80u) list1[pos++] = e0;
                if (pmask & 0x8000u) list1[pos++] = e0 + 1;
                if (pmask & 0x800000u) list1[pos++] = e0 + 2;
                if (pmask & 0x80000000u) list1[pos] = e0 + 3;
            }
        }
        __syncthreads();
        // ---- phase 2: cornerScore of every survivor.  score = (largest t for which the pixel is still a FAST-9 corner), so
        // "corner at T" <=> score >= T and no separate ring test is needed.  A 9-arc is three 3-arcs: min3 / max3 twice.
        const int n1 = s_cnt1;
        for (int i0 = 0; i0 < n1; i0 += NT) {
            const int i = i0 + tid;
            bool corner = false;
            unsigned short ent = 0;
            if (i < n1) {
                ent = list1[i];
                const int sx = ent & 255, sy = ent >> 8;
                const uint8_t* p = tile + (sy + 3) * SD_FS_TW + sx + 4;
                const int S = SD_FS_TW;
                const int v = p[0];
                // Polarity first: a 9-arc holds at least two of the four compass points, so the pixel can only be a "darker ring"
                // corner at T if two compass points are < v - T and a "brighter ring" corner if two are > v + T.  The 16 ring
                // differences are formed with the sign of the possible polarity folded in (one v_mad_i32_i24 each, the price of the
                // plain subtraction), and only that polarity's 9-arc minima are evaluated: half the min3/max3 work.  Both at once
                // (0.2 - 0.7 % of the survivors) is handled by a second evaluation in the waves that hold such a pixel; a polarity
                // that cannot reach T contributes nothing to max(dark, bright) of a corner, so the stored score is unchanged.
                const int r0 = p[3 * S], r4 = p[3], r8 = p[-3 * S], r12 = p[-3];
                // "two compass points darker than v - T" <=> the SECOND SMALLEST of the four is; "two brighter than v + T" <=> the second largest is
                // (six min / max for both, no compare-and-count chains)
                const int m1 = min(r0, r4), M1 = max(r0, r4), m2 = min(r8, r12), M2 = max(r8, r12);
                const int second_lo = min(max(m1, m2), min(M1, M2)), second_hi = max(min(M1, M2), max(m1, m2));
                const bool dark = second_lo < v - T;
                const bool both = dark && second_hi > v + T;
                auto side = [&](const int sg) -> int {
                    const int sv = sg * v, ng = -sg;
                    int d[16];
                    d[0] = __mul24(r0, ng) + sv;             d[1] = __mul24(p[3 * S + 1], ng) + sv;   d[2] = __mul24(p[2 * S + 2], ng) + sv;   d[3] = __mul24(p[S + 3], ng) + sv;
                    d[4] = __mul24(r4, ng) + sv;             d[5] = __mul24(p[-S + 3], ng) + sv;      d[6] = __mul24(p[-2 * S + 2], ng) + sv;  d[7] = __mul24(p[-3 * S + 1], ng) + sv;
                    d[8] = __mul24(r8, ng) + sv;             d[9] = __mul24(p[-3 * S - 1], ng) + sv;  d[10] = __mul24(p[-2 * S - 2], ng) + sv; d[11] = __mul24(p[-S - 3], ng) + sv;
                    d[12] = __mul24(r12, ng) + sv;           d[13] = __mul24(p[S - 3], ng) + sv;      d[14] = __mul24(p[2 * S - 2], ng) + sv;  d[15] = __mul24(p[3 * S - 1], ng) + sv;
                    int mn3[16], a9[16];
#pragma unroll
                    for (int k = 0; k < 16; k++) mn3[k] = min(min(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
#pragma unroll
                    for (int k = 0; k < 16; k++) a9[k] = min(min(mn3[k], mn3[(k + 3) & 15]), mn3[(k + 6) & 15]);
                    a9[0] = max(max(a9[0], a9[1]), a9[2]);    a9[3] = max(max(a9[3], a9[4]), a9[5]);    a9[6] = max(max(a9[6], a9[7]), a9[8]);
                    a9[9] = max(max(a9[9], a9[10]), a9[11]);  a9[12] = max(max(a9[12], a9[13]), a9[14]);
                    return max(max(max(a9[0], a9[3]), a9[6]), max(max(a9[9], a9[12]), a9[15]));
                };
                int sc = side(dark ? 1 : -1) - 1;
                // the other polarity only where the first one did not make the pixel a corner (a pixel is a corner in at most one polarity: 9 + 9 > 16)
                const bool again = both && sc < T;
                if (__any(again)) { if (again) sc = side(-1) - 1; }
                corner = sc >= T;
                if (corner) {
                    score[(sy + 1) * SD_FS_SW + sx + 1] = (uint8_t)sc;
                    list2[SD_FAST_ADD1(&s_cnt2)] = ent;       // one LDS atomic per corner (see phase 1)
                }
            }
        }
        __syncthreads();
        // ---- phase 3: 3x3 strict-maximum NMS over the corner list (zero frame = outside the scanned area); survivors go to a
        //      compact list and set their bit in the row masks
        const int n2 = s_cnt2;
        for (int i0 = 0; i0 < n2; i0 += NT) {
            const int i = i0 + tid;
            bool ok = false;
            unsigned key = 0;
            if (i < n2) {
                const unsigned short ent = list2[i];
                const int sx = ent & 255, sy = ent >> 8;
                const uint8_t* q = score + (sy + 1) * SD_FS_SW + sx + 1;
                const int s = q[0];
                ok = s > q[-SD_FS_SW - 1] && s > q[-SD_FS_SW] && s > q[-SD_FS_SW + 1] && s > q[-1] && s > q[1] &&
                     s > q[SD_FS_SW - 1] && s > q[SD_FS_SW] && s > q[SD_FS_SW + 1];
                key = ((unsigned)ent << 8) | (unsigned)s;
                if (ok) { atomicOr(&rowAll[sy], 1ull << sx); kept[SD_FAST_ADD1(&s_cnt3)] = key; }
            }
        }
        __syncthreads();
        n3 = s_cnt3;
        if (n3 > 0 || pass == 1 || A.iniTh == A.minTh) break;
        if (tid == 0) { s_cnt1 = 0; s_cnt2 = 0; }          // nothing survived at iniTh: once more at minTh (the score tile keeps its, identical, values)
        __syncthreads();
        }
        // ---- phase 4: row-major emission.  rank of a survivor = survivors in the rows above (wave prefix sum over the row
        //      masks, one row per lane) + survivors to its left in its own row (popcount of the masked row word)
        const unsigned long long* rows = rowAll;
        if (tid >= 64 && tid - lane >= n3) return;     // a wave without a survivor of its own has nothing to emit (wave 0 writes the count; n3 <= 64 almost always)
        const int rowPop = lane < SD_FS_ROWS ? __popcll(rows[lane]) : 0;
        // inclusive wave scan on the DPP network (row_shr 1, 2, 4, 8, then lane 15 / 31 of the rows below broadcast): six vector instructions
        // where six __shfl_up steps were thirty (a bpermute through the LDS unit, a compare and a select each)
        int incl = rowPop;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);
        const int excl = incl - rowPop;
        uint32_t* out = cellList + (size_t)img * A.cellListCap + c.listOffset;
        for (int i0 = 0; i0 < n3; i0 += NT) {
            const int i = i0 + tid;
            const unsigned key = i < n3 ? kept[i] : 0u;
            const int sx = (key >> 8) & 255, sy = key >> 16;
            const int base = __shfl(excl, sy, 64);
            if (i < n3) {
                const int rank = base + __popcll(rows[sy] & ((1ull << sx) - 1ull));
                const uint32_t px = (uint32_t)(sx + 3 + c.jw), py = (uint32_t)(sy + 3 + c.ih);
                if (rank < c.cap) out[rank] = px | (py << 12) | ((key & 255u) << 24);
            }
        }
        if (tid == 63) cellCount[(size_t)img * A.cellTotal + ci] = min(incl, c.cap);
    }
}

// ------------------------------------------------------------------ Gaussian blur 7x7, wide-access version
// Tile = 128 x SD_BLUR_TR output pixels per 256-thread workgroup.  Horizontal pass straight from HBM with aligned
// dword loads (the pyramid's own REFLECT_101 frame supplies the halo): per output pixel two
// v_dot4_u32_u8 against the packed taps; exact 8.8 sums go to LDS as packed u16.  Vertical pass from
// LDS (ds_read_b64), 7 mads per pixel, (sum + 0x8000) >> 16, one dword store per 4 pixels.
#ifndef SD_BLUR_TR
#define SD_BLUR_TR 64        // output rows per tile: (TR + 6) / TR rows are filtered horizontally (16: 1.375x, 0.42 ms per 256 images; 32: 0.36; 64: 0.345)
#endif
template <bool CLAMP>
__global__ void __launch_bounds__(256) k_blur_wide(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                                   const SdDevPlan* __restrict__ PP, const int* __restrict__ tiles, int nTiles, int nImages)
{
    const SdDevPlan& P = *PP;
    __shared__ uint2 hbuf[SD_BLUR_TR + 6][32];
    // XCD-aware order over an exact tile list (level | tx << 8 | ty << 16 per tile of one image): all tiles of an image meet
    // in one L2, so the 6 halo rows and the 128-byte lines that neighbouring tiles share are fetched from HBM once.
    int img, item;
    if (!sd_xcd_image_item(blockIdx.x, nTiles, nImages, img, item)) return;
    const int td = tiles[item];
    const int level = td & 255;
    const SdLevel& g = P.lv[level];
    const int x0 = ((td >> 8) & 255) * 128, y0 = (td >> 16) * SD_BLUR_TR;
    const int tid = threadIdx.x;
    const uint32_t tapsLo = (uint32_t)P.taps[0] | ((uint32_t)P.taps[1] << 8) | ((uint32_t)P.taps[2] << 16) | ((uint32_t)P.taps[3] << 24);
    const uint32_t tapsHi = (uint32_t)P.taps[4] | ((uint32_t)P.taps[5] << 8) | ((uint32_t)P.taps[6] << 16);
    const uint8_t* src = pyr + (size_t)img * P.pyrImageBytes + g.pyrOffset + (size_t)SD_EDGE * g.stride + SD_XOFF;
    {
        // horizontal pass: all row requests of a thread go out before the first one is consumed
        constexpr int NIT = (SD_BLUR_TR + 6 + 7) / 8;
        const int gq = tid & 31, rb = tid >> 5;
        const uint8_t* colp = src + x0 + 4 * gq - 4;
        uint32_t w[NIT][3];
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int r = rb + 8 * i;
            const int y = min(y0 + r - 3, g.H + 2);
            const uint32_t* rowp = (const uint32_t*)(colp + (ptrdiff_t)__mul24(y, g.stride));
            if (r < SD_BLUR_TR + 6) { w[i][0] = rowp[0]; w[i][1] = rowp[1]; w[i][2] = rowp[2]; }
        }
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int r = rb + 8 * i;
            if (r < SD_BLUR_TR + 6) {
                uint32_t h[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t A = j == 3 ? w[i][1] : __builtin_amdgcn_alignbyte(w[i][1], w[i][0], j + 1);
                    const uint32_t B = j == 3 ? w[i][2] : __builtin_amdgcn_alignbyte(w[i][2], w[i][1], j + 1);
                    h[j] = __builtin_amdgcn_udot4(A, tapsLo, __builtin_amdgcn_udot4(B, tapsHi, 0u, false), false);
                }
                hbuf[r][gq] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            }
        }
    }
    __syncthreads();
    // Vertical pass: a thread owns one 4-px column group and SD_BLUR_TR / 8 consecutive output rows, and slides a 7-row window
    // of unpacked 8.8 sums down them: one ds_read_b64 and four unpacks per output row instead of seven of each.  The
    // accumulators start at the rounding constant; with taps summing to <= 256 the result is byte 2 of the sum (CLAMP covers
    // the 257 case sd_extractor_set_blur_taps admits).
    uint8_t* dst = blur + (size_t)img * P.blurImageBytes + g.blurOffset;
    constexpr int RPT = SD_BLUR_TR / 8;                   // output rows per thread
    const int gq = tid & 31, r0 = (tid >> 5) * RPT;
    const int x = x0 + 4 * gq;
    if (x >= g.W) return;
    // taps as u16 pairs for v_dot2_u32_u16: (t0,t1) (t2,t3) (t4,t5) and (0,t6); each accumulation step handles two rows of one
    // pixel, so an output pixel costs 4 instructions instead of 7 multiply-adds (+ one v_perm per pixel to pair the incoming row
    // with its predecessor) — the kernel is bound by vector instruction issue
    typedef unsigned short sd_us2 __attribute__((ext_vector_type(2)));
    const sd_us2 t01 = __builtin_bit_cast(sd_us2, (uint32_t)P.taps[0] | ((uint32_t)P.taps[1] << 16));
    const sd_us2 t23 = __builtin_bit_cast(sd_us2, (uint32_t)P.taps[2] | ((uint32_t)P.taps[3] << 16));
    const sd_us2 t45 = __builtin_bit_cast(sd_us2, (uint32_t)P.taps[4] | ((uint32_t)P.taps[5] << 16));
    const sd_us2 t_6 = __builtin_bit_cast(sd_us2, (uint32_t)P.taps[6] << 16);
    // pr[a % 6][j] = (h[row a][column j], h[row a + 1][column j]) as a u16 pair; rows are relative to r0
    uint32_t pr[6][4];
    uint2 prev = hbuf[r0][gq];
#pragma unroll
    for (int a = 0; a < 5; a++) {
        const uint2 cur = hbuf[r0 + a + 1][gq];
        pr[a][0] = __builtin_amdgcn_perm(cur.x, prev.x, 0x05040100u); pr[a][1] = __builtin_amdgcn_perm(cur.x, prev.x, 0x07060302u);
        pr[a][2] = __builtin_amdgcn_perm(cur.y, prev.y, 0x05040100u); pr[a][3] = __builtin_amdgcn_perm(cur.y, prev.y, 0x07060302u);
        prev = cur;
    }
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        const uint2 cur = hbuf[r0 + r + 6][gq];
        uint32_t* np = pr[(r + 5) % 6];
        np[0] = __builtin_amdgcn_perm(cur.x, prev.x, 0x05040100u); np[1] = __builtin_amdgcn_perm(cur.x, prev.x, 0x07060302u);
        np[2] = __builtin_amdgcn_perm(cur.y, prev.y, 0x05040100u); np[3] = __builtin_amdgcn_perm(cur.y, prev.y, 0x07060302u);
        prev = cur;
        uint32_t sum[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t acc = __builtin_amdgcn_udot2(__builtin_bit_cast(sd_us2, pr[r % 6][j]), t01, 0x8000u, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(sd_us2, pr[(r + 2) % 6][j]), t23, acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(sd_us2, pr[(r + 4) % 6][j]), t45, acc, false);
            sum[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(sd_us2, pr[(r + 5) % 6][j]), t_6, acc, false);
        }
        const int y = y0 + r0 + r;
        if (y < g.H) {
            uint32_t o;
            if (CLAMP)
                o = min(sum[0] >> 16, 255u) | (min(sum[1] >> 16, 255u) << 8) | (min(sum[2] >> 16, 255u) << 16) | (min(sum[3] >> 16, 255u) << 24);
            else
                o = __builtin_amdgcn_perm(__builtin_amdgcn_perm(sum[3], sum[2], 0x0C0C0602u), __builtin_amdgcn_perm(sum[1], sum[0], 0x0C0C0602u), 0x05040100u);
            *(uint32_t*)(dst + (size_t)y * g.blurStride + x) = o;
        }
    }
}

// ------------------------------------------------------------------ cvtColor -> gray, 16 px per thread: three unaligned
// 16-byte loads (48 bytes = 16 pixels x 3 channels) and one 16-byte store; rows of 1241 pixels are never aligned, the
// hardware takes unaligned dwordx4 accesses.
__device__ __forceinline__ uint32_t sd_gray4(uint32_t a, uint32_t b, uint32_t c, int cr, int cb)
{
    // 12 bytes = 4 pixels x 3 channels; pixel k: bytes 3k, 3k+1, 3k+2
    const uint32_t p0 = (__umul24(a & 255, cr) + __umul24((a >> 8) & 255, 9617) + __umul24((a >> 16) & 255, cb) + 8192) >> 14;
    const uint32_t p1 = (__umul24(a >> 24, cr) + __umul24(b & 255, 9617) + __umul24((b >> 8) & 255, cb) + 8192) >> 14;
    const uint32_t p2 = (__umul24((b >> 16) & 255, cr) + __umul24(b >> 24, 9617) + __umul24(c & 255, cb) + 8192) >> 14;
    const uint32_t p3 = (__umul24((c >> 8) & 255, cr) + __umul24((c >> 16) & 255, 9617) + __umul24(c >> 24, cb) + 8192) >> 14;
    return p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
}
__global__ void __launch_bounds__(256) k_cvt_gray3_wide(const uint8_t* __restrict__ src, int W, int H, size_t sstride,
                                                        size_t spitch, int rgbOrder, uint8_t* __restrict__ dst,
                                                        size_t dstride, size_t dpitch)
{
    const int img = blockIdx.z;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 16;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (y >= H || x0 >= W) return;
    const uint8_t* srow = src + (size_t)img * spitch + (size_t)y * sstride + (size_t)x0 * 3;
    uint8_t* drow = dst + (size_t)img * dpitch + (size_t)y * dstride + x0;
    const int cr = rgbOrder ? 4899 : 1868, cb = rgbOrder ? 1868 : 4899;
    if (x0 + 15 < W) {
        const sd_u128_unaligned* s = (const sd_u128_unaligned*)srow;
        const sd_u4v a = s[0], b = s[1], c = s[2];
        sd_u4v o;
        o.x = sd_gray4(a.x, a.y, a.z, cr, cb); o.y = sd_gray4(a.w, b.x, b.y, cr, cb);
        o.z = sd_gray4(b.z, b.w, c.x, cr, cb); o.w = sd_gray4(c.y, c.z, c.w, cr, cb);
        *(sd_u128_unaligned*)drow = o;
        return;
    }
    for (int k = 0; k < W - x0; k++) {      // ragged tail of the row: byte path
        const uint8_t* p = srow + 3 * k;
        const int r = rgbOrder ? p[0] : p[2], gg = p[1], b = rgbOrder ? p[2] : p[0];
        drow[k] = (uint8_t)((r * 4899 + gg * 9617 + b * 1868 + (1 << 13)) >> 14);
    }
}

// ------------------------------------------------------------------ cvtColor + pyramid level 0 in one pass
// GrabImageRGBD / GrabImageStereo convert to gray and hand the gray image to the extractor, whose first step copies it into
// the padded level-0 plane (ORBextractor.cc:1127-1128).  For 3-channel input already in HBM the two steps are one pass: the
// gray image is never written and re-read (it IS the interior of level 0).  The arithmetic is sd_gray4's.
//   k_pyr_level0_rgb        all padded rows, the 16-byte groups that lie fully inside the interior: items flattened over
//                           (row, group) so that every wave is full; three unaligned 16-byte loads, one aligned 16-byte store
//   k_pyr_level0_rgb_frame  the four groups per row that touch the reflected frame (X < 0 or X >= W): per-byte path
__global__ void __launch_bounds__(256) k_pyr_level0_rgb(const uint8_t* __restrict__ src, size_t sstride, size_t spitch, int rgbOrder,
                                                        uint8_t* __restrict__ pyr, const SdDevPlan* __restrict__ PP, int groupsPerRow,
                                                        uint32_t gprInv)
{
    const SdDevPlan& P = *PP;
    const SdLevel& g = P.lv[0];
    const int img = blockIdx.y;
    const uint32_t item = blockIdx.x * 256u + threadIdx.x;
    const int Yp = (int)__umulhi(item, gprInv);            // item / groupsPerRow
    if (Yp >= g.H + 2 * SD_EDGE) return;
    const int X0 = 16 * (int)(item - (uint32_t)Yp * (uint32_t)groupsPerRow);          // interior groups start at column 0
    const int sy = sd_reflect101(Yp - SD_EDGE, g.H);
    const uint8_t* srow = src + (size_t)img * spitch + (size_t)sy * sstride;
    const int cr = rgbOrder ? 4899 : 1868, cb = rgbOrder ? 1868 : 4899;
    const sd_u128_unaligned* s = (const sd_u128_unaligned*)(srow + 3 * X0);
    const sd_u4v a = s[0], b = s[1], c = s[2];
    sd_u4v o;
    o.x = sd_gray4(a.x, a.y, a.z, cr, cb); o.y = sd_gray4(a.w, b.x, b.y, cr, cb);
    o.z = sd_gray4(b.z, b.w, c.x, cr, cb); o.w = sd_gray4(c.y, c.z, c.w, cr, cb);
    uint8_t* drow = pyr + (size_t)img * P.pyrImageBytes + g.pyrOffset + (size_t)Yp * g.stride;
    *(sd_u4v*)(drow + SD_XOFF + X0) = o;
}

// 4-channel input (CV_RGBA2GRAY / CV_BGRA2GRAY, Tracking.cc:187-200): 16 pixels = four aligned-or-not 16-byte loads, pixel k = word k,
// bytes 0..2 weighted as above, the alpha byte ignored.
__device__ __forceinline__ uint32_t sd_gray4x4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, int cr, int cb)
{
    const uint32_t p0 = (__umul24(a & 255, cr) + __umul24((a >> 8) & 255, 9617) + __umul24((a >> 16) & 255, cb) + 8192) >> 14;
    const uint32_t p1 = (__umul24(b & 255, cr) + __umul24((b >> 8) & 255, 9617) + __umul24((b >> 16) & 255, cb) + 8192) >> 14;
    const uint32_t p2 = (__umul24(c & 255, cr) + __umul24((c >> 8) & 255, 9617) + __umul24((c >> 16) & 255, cb) + 8192) >> 14;
    const uint32_t p3 = (__umul24(d & 255, cr) + __umul24((d >> 8) & 255, 9617) + __umul24((d >> 16) & 255, cb) + 8192) >> 14;
    return p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
}
__global__ void __launch_bounds__(256) k_pyr_level0_rgba(const uint8_t* __restrict__ src, size_t sstride, size_t spitch, int rgbOrder,
                                                         uint8_t* __restrict__ pyr, const SdDevPlan* __restrict__ PP, int groupsPerRow,
                                                         uint32_t gprInv)
{
    const SdDevPlan& P = *PP;
    const SdLevel& g = P.lv[0];
    const int img = blockIdx.y;
    const uint32_t item = blockIdx.x * 256u + threadIdx.x;
    const int Yp = (int)__umulhi(item, gprInv);            // item / groupsPerRow
    if (Yp >= g.H + 2 * SD_EDGE) return;
    const int X0 = 16 * (int)(item - (uint32_t)Yp * (uint32_t)groupsPerRow);
    const int sy = sd_reflect101(Yp - SD_EDGE, g.H);
    const uint8_t* srow = src + (size_t)img * spitch + (size_t)sy * sstride;
    const int cr = rgbOrder ? 4899 : 1868, cb = rgbOrder ? 1868 : 4899;
    const sd_u128_unaligned* s = (const sd_u128_unaligned*)(srow + 4 * X0);
    const sd_u4v a = s[0], b = s[1], c = s[2], d = s[3];
    sd_u4v o;
    o.x = sd_gray4x4(a.x, a.y, a.z, a.w, cr, cb); o.y = sd_gray4x4(b.x, b.y, b.z, b.w, cr, cb);
    o.z = sd_gray4x4(c.x, c.y, c.z, c.w, cr, cb); o.w = sd_gray4x4(d.x, d.y, d.z, d.w, cr, cb);
    uint8_t* drow = pyr + (size_t)img * P.pyrImageBytes + g.pyrOffset + (size_t)Yp * g.stride;
    *(sd_u4v*)(drow + SD_XOFF + X0) = o;
}

__global__ void __launch_bounds__(256) k_pyr_level0_rgb_frame(const uint8_t* __restrict__ src, size_t sstride, size_t spitch, int rgbOrder,
                                                              uint8_t* __restrict__ pyr, const SdDevPlan* __restrict__ PP, int groupsPerRow,
                                                              int tailGroups, int bpp)
{
    const SdDevPlan& P = *PP;
    const SdLevel& g = P.lv[0];
    const int img = blockIdx.y;
    const int per = 2 + tailGroups;                         // frame groups per row: two on the left (X0 = -32, -16), the rest on the right
    const int item = blockIdx.x * 256 + threadIdx.x;
    const int Yp = item / per, q = item - Yp * per;
    if (Yp >= g.H + 2 * SD_EDGE) return;
    const int X0 = q < 2 ? -SD_XOFF + 16 * q : 16 * (groupsPerRow + q - 2);
    const int sy = sd_reflect101(Yp - SD_EDGE, g.H);
    const uint8_t* srow = src + (size_t)img * spitch + (size_t)sy * sstride;
    const int cr = rgbOrder ? 4899 : 1868, cb = rgbOrder ? 1868 : 4899;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int X = X0 + k;
        X = X < -SD_EDGE ? -SD_EDGE : (X > g.W + SD_EDGE - 1 ? g.W + SD_EDGE - 1 : X);     // margin bytes: any value
        const uint8_t* p = srow + bpp * sd_reflect101(X, g.W);
        const uint32_t v = (__umul24((uint32_t)p[0], (uint32_t)cr) + __umul24((uint32_t)p[1], 9617u) + __umul24((uint32_t)p[2], (uint32_t)cb) + 8192u) >> 14;
        w[k >> 2] |= v << (8 * (k & 3));
    }
    sd_u4v o;
    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
    uint8_t* drow = pyr + (size_t)img * P.pyrImageBytes + g.pyrOffset + (size_t)Yp * g.stride;
    *(sd_u4v*)(drow + SD_XOFF + X0) = o;
}

// The same pair for 8-bit gray input (the stereo path: GrabImageStereo hands two gray images to the extractor): one unaligned
// 16-byte load and one aligned 16-byte store per interior group; the frame groups copy reflected bytes.
__global__ void __launch_bounds__(256) k_pyr_level0_gray(const uint8_t* __restrict__ src, size_t sstride, size_t spitch,
                                                         uint8_t* __restrict__ pyr, const SdDevPlan* __restrict__ PP, int groupsPerRow,
                                                         uint32_t gprInv)
{
    const SdDevPlan& P = *PP;
    const SdLevel& g = P.lv[0];
    const int img = blockIdx.y;
    const uint32_t item = blockIdx.x * 256u + threadIdx.x;
    const int Yp = (int)__umulhi(item, gprInv);            // item / groupsPerRow
    if (Yp >= g.H + 2 * SD_EDGE) return;
    const int X0 = 16 * (int)(item - (uint32_t)Yp * (uint32_t)groupsPerRow);
    const int sy = sd_reflect101(Yp - SD_EDGE, g.H);
    const sd_u4v v = *(const sd_u128_unaligned*)(src + (size_t)img * spitch + (size_t)sy * sstride + X0);
    *(sd_u4v*)(pyr + (size_t)img * P.pyrImageBytes + g.pyrOffset + (size_t)Yp * g.stride + SD_XOFF + X0) = v;
}

__global__ void __launch_bounds__(256) k_pyr_level0_gray_frame(const uint8_t* __restrict__ src, size_t sstride, size_t spitch,
                                                               uint8_t* __restrict__ pyr, const SdDevPlan* __restrict__ PP, int groupsPerRow,
                                                               int tailGroups)
{
    const SdDevPlan& P = *PP;
    const SdLevel& g = P.lv[0];
    const int img = blockIdx.y;
    const int per = 2 + tailGroups;
    const int item = blockIdx.x * 256 + threadIdx.x;
    const int Yp = item / per, q = item - Yp * per;
    if (Yp >= g.H + 2 * SD_EDGE) return;
    const int X0 = q < 2 ? -SD_XOFF + 16 * q : 16 * (groupsPerRow + q - 2);
    const int sy = sd_reflect101(Yp - SD_EDGE, g.H);
    const uint8_t* srow = src + (size_t)img * spitch + (size_t)sy * sstride;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int X = X0 + k;
        X = X < -SD_EDGE ? -SD_EDGE : (X > g.W + SD_EDGE - 1 ? g.W + SD_EDGE - 1 : X);     // margin bytes: any value
        w[k >> 2] |= (uint32_t)srow[sd_reflect101(X, g.W)] << (8 * (k & 3));
    }
    sd_u4v o;
    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
    *(sd_u4v*)(pyr + (size_t)img * P.pyrImageBytes + g.pyrOffset + (size_t)Yp * g.stride + SD_XOFF + X0) = o;
}
