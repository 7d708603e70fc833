/* tests/c/inmemory_key.c -- the layout-independent key upload a Rust binding uses, driven from plain C.
 *
 * plugins/arkworks-mi355x/src/lib.rs::ProvingContext::new hands the five query vectors of an ark_groth16::ProvingKey<E> to the backend as they lie in memory:
 *     zl_bases_upload(ctx, curve, group, query.as_ptr(), len, size_of::<GroupAffine<P>>(), offset_of!(GroupAffine<P>, infinity), ZL_MONT, &handle)
 * i.e. records { x, y: Montgomery limbs; infinity: bool; padding } at a stride that is NOT 2 (or 4) field elements.  Rust cannot be compiled in this image, so this
 * program plays the caller: it compiles a small Poseidon circuit with the library, downloads the five queries (canonical, packed), rebuilds them as such records
 * (Montgomery form computed here by 64 L doublings mod q, infinity as arkworks' (0, 1, true)), uploads those through the stride / offset / ZL_MONT path for G1 AND G2,
 * assembles a zl_g16_pk from the new handles and proves through it: the proof must equal the one made with the library's own key byte for byte, and verify.
 * Built and run by tests/test_abi.py (gcc, -lzl_backend); exit code 0 + "OK" on success. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zl_backend.h"
#include "zl_backend_ext.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, zl_strerror(rc_)); return 1; } } while (0)

/* base-field moduli (little-endian u64 limbs) */
static const uint64_t Q_BLS[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const uint64_t Q_BN[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};

static int geq(const uint64_t* a, const uint64_t* b, int L) {
    for (int i = L - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i];
    return 1;
}
static void sub(uint64_t* a, const uint64_t* b, int L) {
    unsigned __int128 borrow = 0;
    for (int i = 0; i < L; i++) {
        unsigned __int128 d = (unsigned __int128)a[i] - b[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}
/* x <- x * 2^(64 L) mod q: arkworks' in-memory Montgomery form of the canonical integer x */
static void to_mont(uint64_t* x, const uint64_t* q, int L) {
    for (int k = 0; k < 64 * L; k++) {
        uint64_t carry = 0;
        for (int i = 0; i < L; i++) {
            const uint64_t c = x[i] >> 63;
            x[i] = (x[i] << 1) | carry;
            carry = c;
        }
        if (carry || geq(x, q, L)) sub(x, q, L);  /* q < 2^(64 L - 1) for both curves: one subtraction restores x < q */
    }
}

/* `count` packed canonical points (coords field elements of L limbs each; all-zero = infinity) -> records {coords x L limbs Montgomery, u8 infinity, pad} of `stride` bytes */
static unsigned char* make_records(const uint64_t* xy, size_t count, int coords, int L, const uint64_t* q, size_t stride, size_t inf_off) {
    unsigned char* rec = (unsigned char*)malloc(stride * (count ? count : 1));
    memset(rec, 0xA5, stride * (count ? count : 1)); /* padding bytes are garbage, as in a Rust struct */
    for (size_t i = 0; i < count; i++) {
        uint64_t* dst = (uint64_t*)(rec + stride * i);
        const uint64_t* src = xy + (size_t)coords * L * i;
        int inf = 1;
        for (int k = 0; k < coords * L; k++) if (src[k]) inf = 0;
        if (inf) { /* GroupAffine::zero() = (0, 1, true) */
            memset(dst, 0, (size_t)coords * L * 8);
            dst[(coords / 2) * L] = 1; /* y (.c0) = 1 ... */
            to_mont(dst + (coords / 2) * L, q, L); /* ... in Montgomery form */
        } else {
            memcpy(dst, src, (size_t)coords * L * 8);
            for (int c = 0; c < coords; c++) to_mont(dst + c * L, q, L);
        }
        rec[stride * i + inf_off] = (unsigned char)inf;
    }
    return rec;
}

int main(int argc, char** argv) {
    const zl_curve_t curve = (argc > 1 && strcmp(argv[1], "bn254") == 0) ? ZL_BN254 : ZL_BLS12_381;
    const int L = curve == ZL_BLS12_381 ? 6 : 4;
    const uint64_t* q = curve == ZL_BLS12_381 ? Q_BLS : Q_BN;
    const uint32_t k = argc > 2 ? (uint32_t)atoi(argv[2]) : 3;
    zl_ctx* ctx = NULL;
    CHECK(zl_ctx_create(&ctx, 0));
    const uint64_t x0[4] = {11, 0, 0, 0}, x1[4] = {22, 0, 0, 0};
    zl_circuit* circ = NULL;
    CHECK(zl_circuit_poseidon_chain(curve, k, x0, x1, &circ));
    zl_g16_keys* keys = NULL;
    CHECK(zl_groth16_compile(ctx, circ, 77, &keys));
    zl_g16_pk pk;
    CHECK(zl_groth16_keys_pk(keys, &pk));
    zl_r1cs view;
    const uint64_t* assignment = NULL;
    CHECK(zl_circuit_export(circ, &view, &assignment));
    const size_t m1 = (size_t)view.n_instance + view.n_witness;
    size_t N = 1;
    while (N < (size_t)view.n_constraints + view.n_instance) N <<= 1;
    /* the five queries: handle, group, point count */
    const uint64_t old_h[5] = {pk.a_query, pk.b_g1_query, pk.h_query, pk.l_query, pk.b_g2_query};
    const zl_group_t grp[5] = {ZL_G1, ZL_G1, ZL_G1, ZL_G1, ZL_G2};
    const size_t cnt[5] = {m1, m1, N - 1, view.n_witness, m1};
    uint64_t new_h[5];
    size_t n_inf = 0;
    for (int j = 0; j < 5; j++) {
        const int coords = grp[j] == ZL_G1 ? 2 : 4;
        uint64_t* xy = (uint64_t*)calloc(cnt[j] ? cnt[j] : 1, (size_t)coords * L * 8);
        CHECK(zl_bases_download(ctx, old_h[j], 0, cnt[j], xy));
        /* Rust: struct GroupAffine { x: Fq|Fq2, y: Fq|Fq2, infinity: bool } -> size rounds up to the alignment of u64 */
        const size_t inf_off = (size_t)coords * L * 8, stride = inf_off + 8;
        unsigned char* rec = make_records(xy, cnt[j], coords, L, q, stride, inf_off);
        for (size_t i = 0; i < cnt[j]; i++) n_inf += rec[stride * i + inf_off];
        CHECK(zl_bases_upload(ctx, curve, grp[j], rec, cnt[j], stride, (long)inf_off, ZL_MONT | ZL_CHECK, &new_h[j]));
        /* the uploaded points are the downloaded ones */
        uint64_t* back = (uint64_t*)calloc(cnt[j] ? cnt[j] : 1, (size_t)coords * L * 8);
        CHECK(zl_bases_download(ctx, new_h[j], 0, cnt[j], back));
        if (memcmp(xy, back, cnt[j] * (size_t)coords * L * 8) != 0) { fprintf(stderr, "query %d: upload(stride, inf_offset, ZL_MONT) != original\n", j); return 1; }
        free(xy); free(rec); free(back);
    }
    zl_g16_pk pk2 = pk; /* the single points stay; the handles are the in-memory uploads */
    pk2.a_query = new_h[0]; pk2.b_g1_query = new_h[1]; pk2.h_query = new_h[2]; pk2.l_query = new_h[3]; pk2.b_g2_query = new_h[4];
    uint64_t r1cs = 0;
    CHECK(zl_r1cs_upload(ctx, curve, &view, &r1cs));
    const uint64_t r[4] = {0x1234567, 5, 0, 0}, s[4] = {0x7654321, 9, 0, 0};
    zl_g16_proof p1, p2, p3;
    memset(&p1, 0, sizeof p1); memset(&p2, 0, sizeof p2); memset(&p3, 0, sizeof p3);
    CHECK(zl_groth16_prove_resident(ctx, &pk, r1cs, assignment, 0, r, s, &p1));
    CHECK(zl_groth16_prove_resident(ctx, &pk2, r1cs, assignment, 0, r, s, &p2));
    CHECK(zl_groth16_prove(ctx, &pk2, &view, assignment, r, s, &p3));
    if (memcmp(&p1, &p2, sizeof p1) != 0 || memcmp(&p1, &p3, sizeof p1) != 0) { fprintf(stderr, "proof through the in-memory key differs\n"); return 1; }
    int ok = 0;
    CHECK(zl_groth16_verify(keys, assignment + 4, view.n_instance - 1, &p2, &ok));
    if (!ok) { fprintf(stderr, "proof does not verify\n"); return 1; }
    for (int j = 0; j < 5; j++) CHECK(zl_bases_free(ctx, new_h[j]));
    CHECK(zl_r1cs_free(ctx, r1cs));
    zl_groth16_keys_free(keys);
    zl_circuit_free(circ);
    zl_ctx_destroy(ctx);
    printf("OK curve=%d k=%u queries: %zu %zu %zu %zu | %zu points, %zu at infinity\n", (int)curve, k, m1, N - 1, (size_t)view.n_witness, m1, 3 * m1 + N - 1 + view.n_witness, n_inf);
    return 0;
}
