// uc_device.h — device-side data layout shared by the HIP translation units.
//
// HBM layout of the sequence DB (both tracks resident for the whole run; 2 bytes/residue, C4 = 3.6 GB
// of 288 GB):  s3[] / sa[] hold letter codes 0..20, every sequence starts on a 16-byte boundary and is
// followed by >= 16 pad bytes (code 20), so per-lane dword/dwordx4 reads never straddle into unmapped
// memory;  off[i] = byte offset of sequence i (u32: < 4 GiB of padded residues), len[i] = its length.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

namespace uc {

// Function attributes (dynamic LDS beyond 64 KB) live per DEVICE: a launcher keeps one bit per device ordinal and sets
// the attribute the first time it launches on that device (two threads racing on one device both set it: harmless).
struct PerDeviceOnce {
    std::atomic<uint64_t> mask{0};
    template <typename F>
    void operator()(F &&f) {
        int d = 0;
        (void)hipGetDevice(&d);
        const uint64_t bit = 1ull << (d & 63);
        if (!(mask.load(std::memory_order_acquire) & bit)) { f(); mask.fetch_or(bit, std::memory_order_release); }
    }
};

struct DeviceDb {
    uint32_t n = 0;
    const uint8_t *s3 = nullptr, *sa = nullptr;
    const uint16_t *lt = nullptr;    // same offsets: 3Di | AA << 8 per residue, SW_PADPACK in the padding (and at lt[-1])
    const uint32_t *off = nullptr;   // n+1
    const uint32_t *len = nullptr;   // n
    const int8_t *S3 = nullptr, *SA = nullptr;   // 21x21 each
    const int8_t *bias = nullptr;    // rule UC-1/B (off: null): per-residue compositional bias of the 3Di track, same offsets as s3
};

// one workgroup of the gapped kernel = one query + a contiguous run of its pairs
struct SwTask { uint32_t q, begin, count; };

// flag on the column output of the packed known-score start pass (MODE 6): one row holds every optimal cell, so the
// mirror of the pair may take the result with the roles swapped (uc_sw_pk_impl.hpp finish_slot, uc_align.hip sm_scatter_kernel)
constexpr int SW_TE_UNIQUE = 1 << 30;

struct SwArgs {
    DeviceDb db;
    const SwTask *tasks;
    const uint32_t *pt;        // target id per pair
    const int32_t *pqe, *pte;  // mode 2: forward end positions (define the reversed prefixes); mode 3: box ends
    const int32_t *pqs = nullptr, *pts = nullptr;   // mode 3 only: box starts
    uint8_t *tbm = nullptr;                         // packed mode 7: traceback-byte matrices ...
    const unsigned long long *tboff = nullptr;      // ... and the byte offset of every pair's matrix
    int tb_band = 0;                                // packed mode 7: half-width W of the stored diagonal band (0 = whole box), see tb_band_of()
    const int32_t *pscore = nullptr;                // packed modes 4/6: the known optimum score per pair
    int32_t *oscore, *oqe, *ote;
    int open, ext;
    // mode 3: what the traceback statistic adds per path step (diagonal step, identical AA on it, first residue of a gap,
    // further gap residues).  Default = (alignment length << 16 | identities); (0,0,1,0) counts the gaps instead.
    uint32_t tb_diag = 0x10000u, tb_ident = 1u, tb_open = 0x10000u, tb_ext = 0x10000u;
};

constexpr int SW_MAX_ROWS = 2048;   // largest single-strip class (G=64, R=32)

// ---- packed MODE 7: which bytes of a box are stored --------------------------------------------------------------------------------------------
// The traceback of a box [qs..qe] x [0..tl) (columns relative to tStart) runs from (qe, tl-1) to (qs, 0): its diagonal i - c moves from qe - (tl-1)
// to qs, and an optimal path strays from the corridor between the two only by paying for it twice (a gap out and a gap back).  MODE 7 therefore stores the
// H bytes of the diagonals [dlo, dhi] = [min - W, max + W] only - r04 stored the whole box, 1 byte per cell, and ran at 1.5 T cells/s against the
// forward pass's 5.1 T because of those stores (profiles/r05/sw_pass_timing_c4_p500.txt).  Granularity = what a lane produces per step: lane g owns query
// rows [g R, g R + R) and is at column st - g in step st, so it touches the band in the steps [sa, sb] = [g (R+1) - dhi, g (R+1) + R - 1 - dlo]; at any step
// the lanes inside the band are consecutive and at most NL = (dhi - dlo + R - 1) / (R + 1) + 1 of them, so a step's row of the matrix is NL x RB bytes and
// lane g's slot in it is g % NL.  NL >= G (short queries, W = 0) means "everything": slot g, every step.  The walk (tb_walk_kernel) uses the same
// numbers; a read outside the band makes it give up on the pair, which is then redone with the whole box stored.
struct TbBand { int nl, dlo, dhi; };
__host__ __device__ inline TbBand tb_band_of(int qs, int qe, int tl, int G, int R, int W) {
    TbBand b;
    const int d0 = qs, d1 = qe - (tl - 1);
    b.dlo = (d0 < d1 ? d0 : d1) - W;
    b.dhi = (d0 < d1 ? d1 : d0) + W;
    const int nl = (b.dhi - b.dlo + R - 1) / (R + 1) + 1;
    b.nl = (W <= 0 || nl >= G) ? G : nl;
    return b;
}

// host launchers (uc_sw.hip / uc_prefilter.hip)
void launch_sw_class(int G, int R, int mode, const SwArgs &a, uint32_t n_tasks, hipStream_t s);
void launch_sw_pk_class(int G, int R, int mode, const SwArgs &a, uint32_t n_tasks, hipStream_t s);
// queries beyond the systolic classes, MODE 0 / 1 / 2 / 3: row-blocked systolic kernel (uc_sw_long.hip); a.tasks = the long-query
// tasks (<= SW_LONG_TASK_PAIRS pairs each), pair_base = plan index of the first long-query pair, work = 2 (MODE 3: 4) x stride ints per pair
constexpr uint32_t SW_LONG_TASK_PAIRS = 8;
size_t sw_long_work_ints(int mode, uint32_t n_pairs, uint32_t max_len, uint32_t *stride);
void launch_sw_long(int mode, const SwArgs &a, uint32_t n_tasks, uint32_t pair_base, int32_t *work, uint32_t stride, hipStream_t s);
void launch_ungapped(const DeviceDb &db, uint64_t n, const uint32_t *q, const uint32_t *t, const int32_t *diag,
                     int32_t *score, unsigned long long *overlap_sum /* nullable */, hipStream_t s);
bool sw_class_for(int lq, int *G, int *R);
// padded device layout of the sequence tracks from the raw (unpadded) ones: s3 / sa[total] with pad letter 20, lt[total + 16]
// (16 PAD pairs in front) with the PAD pair in all padding; off = padded offsets (n + 1), roff = raw offsets (n + 1)
void launch_comp_bias(const DeviceDb &db, int scale_milli, int8_t *out, hipStream_t s);   // rule UC-1/B
void launch_db_pad(uint32_t n, const uint32_t *off, const uint32_t *len, const uint32_t *cur /* nullable: raw sequence id of sequence i */,
                   const uint64_t *roff, const uint8_t *r3, const uint8_t *ra,
                   uint64_t total, uint8_t *s3, uint8_t *sa, uint16_t *lt, hipStream_t s);

}  // namespace uc
