// zl_serialize.h -- arkworks 0.3 `CanonicalSerialize` (compressed) for G1 / G2 points and Groth16 proofs.  Host only.
//
// Replaces what `proof_as_bytes` / `Proof: HasSerialization` reach in the reference
// (/root/reference/plugins/arkworks/src/groth16.rs:68-107 -> ark_groth16::Proof::serialize -> GroupAffine::serialize ->
// Fp::serialize_with_flags with SWFlags).  Restated from the published ark-serialize / ark-ec / ark-ff 0.3 format:
//   point      = x, little-endian canonical integer, ceil((MODULUS_BITS + 2) / 8) bytes per base-field element
//                (48 for BLS12-381 Fq, 32 for BN254 Fq); Fq2: c0 (no flags) then c1 (flags)
//   flag bits  = top two bits of the LAST byte: bit 7 = "y is the larger of {y, -y}" (Fp: canonical integers compared; Fq2: c1 first,
//                then c0), bit 6 = point at infinity (x = 0)
//   proof      = A (G1) || B (G2) || C (G1): 192 bytes for BLS12-381, 128 for BN254
// Decompression solves y^2 = x^3 + b (q = 3 mod 4 for both curves: y = a^((q+1)/4); Fq2 through the norm), picks the root the flag
// names and, like ark-ec's deserializer, rejects points outside the prime-order subgroup.
//   uncompressed = x || y, flags on the last byte of y (bit 6 = infinity, written as (0, 1); a finite point has no flag bits); Vec<T> = u64
//                little-endian length, then the elements; ProvingKey<E> = vk (alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1), beta_g1,
//                delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query in declaration order (ark-groth16 0.3 data_structures.rs) --
//                the reference writes its ProvingContext with serialize_unchecked = this form (groth16.rs:142-179)
// NOT VERIFIED against bytes produced by arkworks: the reference holds no serialized vector and cannot be built here.  The
// tests pin it to an independent Python restatement (oracle/pyoracle.py) and to round trips only.
#pragma once
#include <stddef.h>
#include <string.h>
#include "zl_curve.h"

namespace openzl {
namespace serialize {

template <class FqP, class FrP, class C1, class C2>
struct Codec {
    using F = Fp<FqP>;
    using F2 = Fp2<FqP>;
    static constexpr int NB = FqP::N * 4;  // bytes per Fq element: both moduli leave >= 2 spare bits in their top byte
    static constexpr size_t G1_BYTES = NB, G2_BYTES = 2 * NB, PROOF_BYTES = 4 * NB;

    static F b1() { F r; for (int i = 0; i < F::N; i++) r.l[i] = C1::b(i); return r; }
    static F2 b2() { F2 r; for (int i = 0; i < F::N; i++) { r.c0.l[i] = C2::b0(i); r.c1.l[i] = C2::b1(i); } return r; }
    // canonical integer comparison a > b
    static bool gt(const F& a_mont, const F& b_mont) {
        const F a = zl::from_mont(a_mont), b = zl::from_mont(b_mont);
        for (int i = F::N - 1; i >= 0; i--) {
            if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
        }
        return false;
    }
    static bool gt(const F2& a, const F2& b) {  // ark-ff QuadExtField::cmp: c1 first, then c0
        if (a.c1 != b.c1) return gt(a.c1, b.c1);
        return gt(a.c0, b.c0);
    }
    static bool sqrt(const F& a, F& out) {  // q = 3 mod 4
        uint32_t e[F::N];
        uint64_t carry = 1;
        for (int i = 0; i < F::N; i++) {
            const uint64_t v = (uint64_t)FqP::mod(i) + carry;
            e[i] = (uint32_t)v;
            carry = v >> 32;
        }
        for (int i = 0; i < F::N; i++) e[i] = (e[i] >> 2) | (i + 1 < F::N ? e[i + 1] << 30 : (uint32_t)(carry << 30));
        const F r = zl::pow_words(a, e, 32 * F::N);
        if (zl::sqr(r) != a) return false;
        out = r;
        return true;
    }
    static bool sqrt(const F2& a, F2& out) {  // u^2 = -1: through the norm
        if (a.c1.is_zero()) {
            F r;
            if (sqrt(a.c0, r)) { out = F2{r, F::zero()}; return true; }
            if (sqrt(zl::neg(a.c0), r)) { out = F2{F::zero(), r}; return true; }  // (r u)^2 = -r^2
            return false;
        }
        F s;
        if (!sqrt(zl::add(zl::sqr(a.c0), zl::sqr(a.c1)), s)) return false;
        const F half = zl::inv(zl::from_u64<FqP>(2));
        F x0;
        if (!sqrt(zl::mul(zl::add(a.c0, s), half), x0) && !sqrt(zl::mul(zl::sub(a.c0, s), half), x0)) return false;
        const F x1 = zl::mul(a.c1, zl::inv(zl::dbl(x0)));
        out = F2{x0, x1};
        return zl::sqr(out) == a;
    }
    template <class T>
    static T load_mont(const uint64_t* w);
    static void put_fq(uint8_t* out, const F& mont, uint8_t flags) {
        const F c = zl::from_mont(mont);
        memcpy(out, c.l, NB);
        out[NB - 1] |= flags;
    }
    static bool get_fq(const uint8_t* in, bool strip_flags, F& out) {  // false if the integer is not < q
        F c;
        memcpy(c.l, in, NB);
        if (strip_flags) c.l[F::N - 1] &= 0x3FFFFFFFu;
        for (int i = F::N - 1; i >= 0; i--) {
            if (c.l[i] != FqP::mod(i)) {
                if (c.l[i] > FqP::mod(i)) return false;
                out = zl::to_mont(c);
                return true;
            }
        }
        return false;  // == q
    }
    // xy: canonical affine words as everywhere in the C ABI (all-zero = infinity)
    static void g1_to_bytes(const uint64_t* xy, bool inf, uint8_t* out) {
        memset(out, 0, G1_BYTES);
        if (inf) { out[NB - 1] |= 0x40; return; }
        F x, y;
        memcpy(x.l, xy, NB);
        memcpy(y.l, reinterpret_cast<const uint8_t*>(xy) + NB, NB);
        x = zl::to_mont(x);
        y = zl::to_mont(y);
        put_fq(out, x, gt(y, zl::neg(y)) ? 0x80 : 0x00);
    }
    static void g2_to_bytes(const uint64_t* xy, bool inf, uint8_t* out) {
        memset(out, 0, G2_BYTES);
        if (inf) { out[G2_BYTES - 1] |= 0x40; return; }
        const uint8_t* p = reinterpret_cast<const uint8_t*>(xy);
        F2 x, y;
        memcpy(x.c0.l, p, NB); memcpy(x.c1.l, p + NB, NB); memcpy(y.c0.l, p + 2 * NB, NB); memcpy(y.c1.l, p + 3 * NB, NB);
        x = zl::to_mont(x);
        y = zl::to_mont(y);
        put_fq(out, x.c0, 0);
        put_fq(out + NB, x.c1, gt(y, zl::neg(y)) ? 0x80 : 0x00);
    }
    template <class T>
    static bool in_subgroup(const T& x, const T& y) {
        uint32_t r[8];
        for (int i = 0; i < 8; i++) r[i] = i < FrP::N ? FrP::mod(i) : 0;
        const XYZZ<T> p{x, y, T::one(), T::one()};
        return zl::mul_scalar(p, r).is_inf();
    }
    // returns ZL_OK / ZL_EINVAL (malformed: non-canonical x, both flags, non-zero x at infinity) / ZL_ENOTCURVE (no y, or outside the subgroup)
    static int g1_from_bytes(const uint8_t* in, uint64_t* xy, uint8_t* inf) {
        const uint8_t flags = in[NB - 1] & 0xC0;
        memset(xy, 0, 2 * NB);
        F x;
        if (flags == 0xC0 || !get_fq(in, true, x)) return ZL_EINVAL;
        if (flags & 0x40) { if (!x.is_zero()) return ZL_EINVAL; *inf = 1; return ZL_OK; }
        F y;
        if (!sqrt(zl::add(zl::mul(zl::sqr(x), x), b1()), y)) return ZL_ENOTCURVE;
        if (gt(y, zl::neg(y)) != ((flags & 0x80) != 0)) y = zl::neg(y);
        if (!in_subgroup(x, y)) return ZL_ENOTCURVE;
        const F xc = zl::from_mont(x), yc = zl::from_mont(y);
        memcpy(xy, xc.l, NB);
        memcpy(reinterpret_cast<uint8_t*>(xy) + NB, yc.l, NB);
        *inf = 0;
        return ZL_OK;
    }
    static int g2_from_bytes(const uint8_t* in, uint64_t* xy, uint8_t* inf) {
        const uint8_t flags = in[G2_BYTES - 1] & 0xC0;
        memset(xy, 0, 4 * NB);
        F2 x;
        if (flags == 0xC0 || !get_fq(in, false, x.c0) || !get_fq(in + NB, true, x.c1)) return ZL_EINVAL;
        if (in[NB - 1] & 0xC0) return ZL_EINVAL;  // c0 carries no flags
        if (flags & 0x40) { if (!x.is_zero()) return ZL_EINVAL; *inf = 1; return ZL_OK; }
        F2 y;
        if (!sqrt(zl::add(zl::mul(zl::sqr(x), x), b2()), y)) return ZL_ENOTCURVE;
        if (gt(y, zl::neg(y)) != ((flags & 0x80) != 0)) y = zl::neg(y);
        if (!in_subgroup(x, y)) return ZL_ENOTCURVE;
        const F2 xc = zl::from_mont(x), yc = zl::from_mont(y);
        uint8_t* p = reinterpret_cast<uint8_t*>(xy);
        memcpy(p, xc.c0.l, NB); memcpy(p + NB, xc.c1.l, NB); memcpy(p + 2 * NB, yc.c0.l, NB); memcpy(p + 3 * NB, yc.c1.l, NB);
        *inf = 0;
        return ZL_OK;
    }
    // ---- uncompressed form (serialize_uncompressed / serialize_unchecked): x then y, SWFlags on the last byte of y.  A finite point carries
    // no flag bits (SWFlags::default() = NegativeY = 0: the sign bit is only meaningful in the compressed form), infinity is
    // GroupAffine::zero() = (0, 1) with bit 6 set.  This is what ProvingContext's codec::Encode writes (groth16.rs:166-179).
    static constexpr size_t G1_UNC_BYTES = 2 * NB, G2_UNC_BYTES = 4 * NB;
    static void g1_to_uncompressed(const uint64_t* xy, bool inf, uint8_t* out) {
        if (inf) { memset(out, 0, G1_UNC_BYTES); out[NB] = 1; out[G1_UNC_BYTES - 1] |= 0x40; return; }
        memcpy(out, xy, G1_UNC_BYTES);  // canonical little-endian words ARE the little-endian bytes
    }
    static void g2_to_uncompressed(const uint64_t* xy, bool inf, uint8_t* out) {
        if (inf) { memset(out, 0, G2_UNC_BYTES); out[2 * NB] = 1; out[G2_UNC_BYTES - 1] |= 0x40; return; }
        memcpy(out, xy, G2_UNC_BYTES);
    }
    // check = false: deserialize_unchecked (coordinates must be canonical integers; nothing else is verified, as in the reference's
    // Decode for ProvingContext, groth16.rs:147-164); check = true additionally requires a point of the prime-order subgroup
    static bool canon_lt_q(const uint32_t* w) {
        for (int i = F::N - 1; i >= 0; i--)
            if (w[i] != FqP::mod(i)) return w[i] < FqP::mod(i);
        return false;
    }
    template <int K>  // K base-field elements: 2 (G1) or 4 (G2)
    static int from_uncompressed(const uint8_t* in, bool check, uint64_t* xy, uint8_t* inf) {
        const uint8_t flags = in[K * NB - 1] & 0xC0;
        uint32_t w[K][F::N];
        for (int i = 0; i < K; i++) {
            memcpy(w[i], in + i * NB, NB);
            if (i + 1 < K) { if (w[i][F::N - 1] >> 30) return ZL_EINVAL; }  // only the last element carries flags
            else w[i][F::N - 1] &= 0x3FFFFFFFu;
            if (!canon_lt_q(w[i])) return ZL_EINVAL;  // ark-ff rejects integers >= q
        }
        memset(xy, 0, K * NB);
        if (flags == 0xC0) return ZL_EINVAL;
        if (flags & 0x40) { *inf = 1; return ZL_OK; }  // coordinates of an infinity record are ignored (GroupAffine::new(x, y, true))
        *inf = 0;
        if (check) {
            F c[K];
            for (int i = 0; i < K; i++) { memcpy(c[i].l, w[i], NB); c[i] = zl::to_mont(c[i]); }
            if constexpr (K == 2) {
                if (zl::sqr(c[1]) != zl::add(zl::mul(zl::sqr(c[0]), c[0]), b1())) return ZL_ENOTCURVE;
                if (!in_subgroup(c[0], c[1])) return ZL_ENOTCURVE;
            } else {
                const F2 x{c[0], c[1]}, y{c[2], c[3]};
                if (zl::sqr(y) != zl::add(zl::mul(zl::sqr(x), x), b2())) return ZL_ENOTCURVE;
                if (!in_subgroup(x, y)) return ZL_ENOTCURVE;
            }
        }
        memcpy(xy, w, K * NB);
        return ZL_OK;
    }
    static int g1_from_uncompressed(const uint8_t* in, bool check, uint64_t* xy, uint8_t* inf) { return from_uncompressed<2>(in, check, xy, inf); }
    static int g2_from_uncompressed(const uint8_t* in, bool check, uint64_t* xy, uint8_t* inf) { return from_uncompressed<4>(in, check, xy, inf); }
};

using BlsCodec = Codec<BLS12_381_Fq, BLS12_381_Fr, BLS12_381_G1, BLS12_381_G2>;
using BnCodec = Codec<BN254_Fq, BN254_Fr, BN254_G1, BN254_G2>;

}  // namespace serialize
}  // namespace openzl
