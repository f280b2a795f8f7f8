// acx_hostshim.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the __host__ __device__ cores of the HIP kernels (ac-solver_amd/csrc/acx_word.h,
// acx_bytes.h) for the CPU so that the `-m "not gpu"` suite can check the exact lane code against the
// oracle without a GPU.  The product never loads this library; the GPU parity tests call the real
// kernels through libacx.so.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../ac-solver_amd/csrc/acx_bytes.h"
#include "../../ac-solver_amd/csrc/acx_word.h"
#include "../../ac-solver_amd/csrc/acx_keys.h"

using namespace acx;

template <typename W>
static void move_packed_rows(const int8_t* in, const uint8_t* act, int64_t n, int L, int cyclical, int8_t* out, int32_t* len, uint8_t* err) {
    for (int64_t r = 0; r < n; r++) {
        const int8_t* row = in + r * 2 * L;
        int8_t* o = out + r * 2 * L;
        memcpy(o, row, 2 * L);
        Pres<W> s;
        bool ok = pack_relator<W>(row, L, s.w0, s.n0);
        ok = pack_relator<W>(row + L, L, s.w1, s.n1) && ok;
        int e;
        if (!ok) {
            e = ACX_ERR_UNPACKABLE;
            s.n0 = s.n1 = 0;
            for (int k = 0; k < L; k++) s.n0 += row[k] != 0, s.n1 += row[L + k] != 0;
        } else if (act[r] >= 12) {
            e = ACX_ERR_ASSERT;
        } else {
            e = apply_move<W>(s, act[r], L, cyclical != 0);
            if (e == ACX_ERR_NONE) {
                unpack_relator<W>(s.w0, s.n0, L, o);
                unpack_relator<W>(s.w1, s.n1, L, o + L);
                // cross-check the 8-letters-at-a-time unpacker against the scalar one
                for (int k0 = 0; k0 < L; k0 += 8) {
                    uint64_t a = unpack8<W>(s.w0, s.n0, k0), b = unpack8<W>(s.w1, s.n1, k0);
                    for (int k = k0; k < k0 + 8 && k < L; k++) {
                        if ((int8_t)(a >> (8 * (k - k0))) != o[k] || (int8_t)(b >> (8 * (k - k0))) != o[L + k]) e = 99;
                    }
                }
            }
        }
        len[2 * r] = s.n0;
        len[2 * r + 1] = s.n1;
        err[r] = (uint8_t)e;
    }
}

// steady-state fast path: rows whose relators are both non-empty and cyclically reduced; others are skipped (err 251)
template <typename W> static void move_reduced_rows(const int8_t* in, const uint8_t* act, int64_t n, int L, int8_t* out, int32_t* len, uint8_t* err) {
    for (int64_t r = 0; r < n; r++) {
        const int8_t* row = in + r * 2 * L;
        int8_t* o = out + r * 2 * L;
        memcpy(o, row, 2 * L);
        Pres<W> s;
        bool ok = pack_relator<W>(row, L, s.w0, s.n0);
        ok = pack_relator<W>(row + L, L, s.w1, s.n1) && ok;
        ok = ok && s.n0 > 0 && s.n1 > 0 && is_cyc_reduced<W>(s.w0, s.n0) && is_cyc_reduced<W>(s.w1, s.n1) && act[r] < 12;
        int e = 251;
        if (ok) {
            e = apply_move_reduced<W>(s, act[r], L);
            if (e == ACX_ERR_NONE) {
                for (int j = 0; 4 * j < L; j++) {  // dword-wise unpack (v_perm path on the device)
                    uint32_t a = relator_dword<W>(s.w0, s.n0, j), b = relator_dword<W>(s.w1, s.n1, j);
                    for (int k = 4 * j; k < 4 * j + 4 && k < L; k++) {
                        o[k] = (int8_t)(a >> (8 * (k - 4 * j)));
                        o[L + k] = (int8_t)(b >> (8 * (k - 4 * j)));
                    }
                }
            }
        }
        len[2 * r] = s.n0;
        len[2 * r + 1] = s.n1;
        err[r] = (uint8_t)e;
    }
}

// search fast path: rows in normal form (is_normal_form); others are skipped (err 251)
template <typename W> static void move_nf_rows(const int8_t* in, const uint8_t* act, int64_t n, int L, int cyclical, int8_t* out, int32_t* len, uint8_t* err) {
    for (int64_t r = 0; r < n; r++) {
        const int8_t* row = in + r * 2 * L;
        int8_t* o = out + r * 2 * L;
        memcpy(o, row, 2 * L);
        Pres<W> s;
        bool ok = pack_relator<W>(row, L, s.w0, s.n0);
        ok = pack_relator<W>(row + L, L, s.w1, s.n1) && ok;
        ok = ok && act[r] < 12 && is_normal_form<W>(s, cyclical != 0);
        int e = 251;
        if (ok) {
            e = apply_move_nf<W>(s, act[r], L, cyclical != 0);
            if (e == ACX_ERR_NONE) {
                unpack_relator<W>(s.w0, s.n0, L, o);
                unpack_relator<W>(s.w1, s.n1, L, o + L);
            }
        }
        len[2 * r] = s.n0;
        len[2 * r + 1] = s.n1;
        err[r] = (uint8_t)e;
    }
}

extern "C" {

// keyops<u128x> (csrc/acx_keys.h: the 128-bit key of a freely reduced word of up to 64 letters): key of the word `letters`, and the word /
// length it decodes back to.  Also a move through Pres<u128x> (the word functions take the type unchanged): shim_move_long.
void shim_long_key(const int8_t* letters, int n, uint64_t* key2, int32_t* n_back, int8_t* letters_back) {
    u128x w;
    int len = 0;
    std::vector<int8_t> row(64, 0);
    for (int k = 0; k < n; k++) row[k] = letters[k];
    (void)pack_relator<u128x>(row.data(), 64, w, len);
    const u128x key = keyops<u128x>::make(w, len);
    key2[0] = (uint64_t)(u128)key;
    key2[1] = (uint64_t)((u128)key >> 64);
    u128x wb;
    int nb;
    keyops<u128x>::split(key, wb, nb);
    *n_back = nb;
    unpack_relator<u128x>(wb, nb, 64, letters_back);
}

void shim_move_long(const int8_t* in, const uint8_t* act, int64_t n, int L, int cyclical, int8_t* out, int32_t* len, uint8_t* err) {
    move_packed_rows<u128x>(in, act, n, L, cyclical, out, len, err);
}

void shim_move_nf(const int8_t* in, const uint8_t* act, int64_t n, int L, int cyclical, int wide, int8_t* out, int32_t* len, uint8_t* err) {
    if (wide) move_nf_rows<u128>(in, act, n, L, cyclical, out, len, err);
    else move_nf_rows<uint64_t>(in, act, n, L, cyclical, out, len, err);
}

void shim_move_reduced(const int8_t* in, const uint8_t* act, int64_t n, int L, int wide, int8_t* out, int32_t* len, uint8_t* err) {
    if (wide) move_reduced_rows<u128>(in, act, n, L, out, len, err);
    else move_reduced_rows<uint64_t>(in, act, n, L, out, len, err);
}

void shim_move_packed(const int8_t* in, const uint8_t* act, int64_t n, int L, int cyclical, int wide, int8_t* out, int32_t* len, uint8_t* err) {
    if (wide) move_packed_rows<u128>(in, act, n, L, cyclical, out, len, err);
    else move_packed_rows<uint64_t>(in, act, n, L, cyclical, out, len, err);
}

void shim_move_bytes(const int8_t* in, const uint8_t* act, int64_t n, int L, int flags, int8_t* out, int32_t* len, uint8_t* err, int32_t* fit) {
    int8_t res[2 * kMaxBytesL], w1[kMaxBytesL], w2[kMaxBytesL];
    for (int64_t r = 0; r < n; r++) {
        const int8_t* row = in + r * 2 * L;
        int lens[2] = {0, 0}, f = -1, ext;
        int a = (flags & ACX_F_NO_MOVE) ? 0 : act[r];
        int e = a >= 12 ? (int)ACX_ERR_ASSERT : move_bytes(row, L, a, flags, res, lens, &f, w1, w2);
        if (e) {
            memcpy(res, row, 2 * L);
            lens[0] = take_nonzero(row, L, w1, &ext);
            lens[1] = take_nonzero(row + L, L, w1, &ext);
            f = -1;
        }
        memcpy(out + r * 2 * L, res, 2 * L);
        len[2 * r] = lens[0];
        len[2 * r + 1] = lens[1];
        err[r] = (uint8_t)e;
        if (fit) fit[r] = f;
    }
}

void shim_simplify_rows(const int8_t* in, int64_t n, int width, int cyclical, int8_t* out, int32_t* len, uint8_t* err) {
    for (int64_t r = 0; r < n; r++) {
        int nz;
        int nn = simplify_row(in + r * width, width, cyclical != 0, out + r * width, &nz);
        if (nn < 0) memcpy(out + r * width, in + r * width, width);
        len[2 * r] = nn < 0 ? nz : nn;
        len[2 * r + 1] = nz;
        err[r] = nn < 0 ? (uint8_t)(-nn) : 0;
    }
}

// signed-tuple comparison of two int8 presentations through the packed comparator
int shim_compare(const int8_t* a, const int8_t* b, int L, int wide) {
    if (wide) {
        Pres<u128> x, y;
        pack_relator<u128>(a, L, x.w0, x.n0); pack_relator<u128>(a + L, L, x.w1, x.n1);
        pack_relator<u128>(b, L, y.w0, y.n0); pack_relator<u128>(b + L, L, y.w1, y.n1);
        return compare_pres<u128>(x, y);
    }
    Pres<uint64_t> x, y;
    pack_relator<uint64_t>(a, L, x.w0, x.n0); pack_relator<uint64_t>(a + L, L, x.w1, x.n1);
    pack_relator<uint64_t>(b, L, y.w0, y.n0); pack_relator<uint64_t>(b + L, L, y.w1, y.n1);
    return compare_pres<uint64_t>(x, y);
}
}
