// zl_pairing.h -- host-side pairing for Groth16::verify (row f4; ms-scale CPU work, not on the accelerated path).
//
// Replaces what `Groth16::<E>::verify` (/root/reference/plugins/arkworks/src/groth16.rs:459-466) delegates to
// ark_groth16::verify_proof_with_prepared_inputs -> E::miller_loop + E::final_exponentiation, and what the plugin's
// PairingEngineExt helpers use (plugins/arkworks/src/pairing.rs:47-90; its tests check bilinearity, :116-129).
// Tower-free formulation: Fq12 = Fq[w]/(w^12 - M6 w^6 + M0_NEG) as 12 Fq coefficients; G2 arithmetic stays on the twist
// in affine Fq2 coordinates (one Fq inversion per step); the line through the untwisted points evaluated at P = (xP, yP) is
//     BN254      (X = x' w^2, Y = y' w^3):  l = -yP + (m' xP) w + (y1' - m' x1') w^3
//     BLS12-381  (X = x'/w^2, Y = y'/w^3):  l * w^3 = (y1' - m' x1') + (m' xP) w^2 - yP w^3
// with m' the Fq2 slope on the twist; the factor w^3 lies in the proper subfield Fq4 and is annihilated by the final
// exponentiation (q^12 - 1)/r (the factor q^6 - 1 by the Frobenius w -> -w and one inversion, the factor (q^6 + 1)/r as one power).  Any non-degenerate bilinear pairing decides Groth16 verification the
// same way; values after the final exponentiation are unique and are compared coefficient-wise with oracle/pyoracle.py.
#pragma once
#include <utility>
#include <vector>
#include "zl_curve.h"

namespace openzl {
namespace pairing {

template <class FqP, class PP>
struct Engine {
    using F = Fp<FqP>;
    using F2 = Fp2<FqP>;
    struct Fq12 { F c[12]; };

    static F small(int v) {  // small non-negative integer -> Montgomery
        return zl::from_u64<FqP>((uint64_t)v);
    }
    static Fq12 one() {
        Fq12 r;
        for (auto& x : r.c) x = F::zero();
        r.c[0] = F::one();
        return r;
    }
    static Fq12 mul(const Fq12& a, const Fq12& b) {
        F t[23];
        for (auto& x : t) x = F::zero();
        for (int i = 0; i < 12; i++) {
            if (a.c[i].is_zero()) continue;
            for (int j = 0; j < 12; j++) {
                if (b.c[j].is_zero()) continue;
                t[i + j] = zl::add(t[i + j], zl::mul(a.c[i], b.c[j]));
            }
        }
        static const F m6 = small(PP::M6), m0 = small(PP::M0_NEG);
        for (int k = 22; k >= 12; k--) {  // w^k = w^(k-12) (M6 w^6 - M0_NEG)
            if (t[k].is_zero()) continue;
            t[k - 6] = zl::add(t[k - 6], zl::mul(t[k], m6));
            t[k - 12] = zl::sub(t[k - 12], zl::mul(t[k], m0));
        }
        Fq12 r;
        for (int i = 0; i < 12; i++) r.c[i] = t[i];
        return r;
    }
    // a^2: 78 products instead of 144 (cross terms once, doubled) -- two thirds of the final exponentiation's work are squarings
    static Fq12 sqr(const Fq12& a) {
        F t[23];
        for (auto& x : t) x = F::zero();
        for (int i = 0; i < 12; i++) {
            if (a.c[i].is_zero()) continue;
            t[2 * i] = zl::add(t[2 * i], zl::sqr(a.c[i]));
            for (int j = i + 1; j < 12; j++) {
                if (a.c[j].is_zero()) continue;
                const F p = zl::mul(a.c[i], a.c[j]);
                t[i + j] = zl::add(t[i + j], zl::add(p, p));
            }
        }
        static const F m6 = small(PP::M6), m0 = small(PP::M0_NEG);
        for (int k = 22; k >= 12; k--) {
            if (t[k].is_zero()) continue;
            t[k - 6] = zl::add(t[k - 6], zl::mul(t[k], m6));
            t[k - 12] = zl::sub(t[k - 12], zl::mul(t[k], m0));
        }
        Fq12 r;
        for (int i = 0; i < 12; i++) r.c[i] = t[i];
        return r;
    }
    static bool eq(const Fq12& a, const Fq12& b) {
        for (int i = 0; i < 12; i++) if (a.c[i] != b.c[i]) return false;
        return true;
    }
    // a0 + a1 i placed at w^k (i = w^6 - ISHIFT): coefficient k gets a0 - ISHIFT a1, coefficient k+6 gets a1
    static void put_fq2(Fq12& l, int k, const F2& a) {
        static const F sh = small(PP::ISHIFT);
        l.c[k] = zl::add(l.c[k], zl::sub(a.c0, zl::mul(sh, a.c1)));
        l.c[k + 6] = zl::add(l.c[k + 6], a.c1);
    }
    static F2 f2_scale(const F2& a, const F& s) { return F2{zl::mul(a.c0, s), zl::mul(a.c1, s)}; }

    struct G2Aff { F2 x, y; bool inf; };
    // line through R and S (R == S: tangent) evaluated at P, and R <- R + S on the twist
    static Fq12 line_and_add(G2Aff& R, const G2Aff& S, const F& xP, const F& yP) {
        Fq12 l;
        for (auto& x : l.c) x = F::zero();
        F2 m;
        bool vertical = false;
        if (R.x == S.x) {
            if (R.y == S.y && !R.y.is_zero()) {
                F2 xx = zl::sqr(R.x);
                m = zl::mul(zl::add(zl::dbl(xx), xx), zl::inv(zl::dbl(R.y)));
            } else {
                vertical = true;
            }
        } else {
            m = zl::mul(zl::sub(S.y, R.y), zl::inv(zl::sub(S.x, R.x)));
        }
        if (vertical) {
            // l = xP - X1 (does not occur for points of prime order r inside the loop; kept for completeness)
            F2 nx = zl::neg(R.x);
            if (PP::TWIST_DIV) { l.c[2] = xP; put_fq2(l, 0, nx); }  // (xP - x1'/w^2) * w^2
            else { l.c[0] = xP; put_fq2(l, 2, nx); }
            R.inf = true;
            return l;
        }
        const F2 mx = f2_scale(m, xP);                    // m' xP
        const F2 c0 = zl::sub(R.y, zl::mul(m, R.x));      // y1' - m' x1'
        if (PP::TWIST_DIV) {
            put_fq2(l, 0, c0);
            put_fq2(l, 2, mx);
            l.c[3] = zl::sub(l.c[3], yP);
        } else {
            l.c[0] = zl::neg(yP);
            put_fq2(l, 1, mx);
            put_fq2(l, 3, c0);
        }
        const F2 x3 = zl::sub(zl::sub(zl::sqr(m), R.x), S.x);
        const F2 y3 = zl::sub(zl::mul(m, zl::sub(R.x, x3)), R.y);
        R.x = x3;
        R.y = y3;
        return l;
    }
    static F2 f2_pow(const F2& a, const uint32_t* e, int nbits) {
        F2 acc = F2::one();
        for (int i = nbits - 1; i >= 0; i--) {
            acc = zl::sqr(acc);
            if ((e[i >> 5] >> (i & 31)) & 1) acc = zl::mul(acc, a);
        }
        return acc;
    }
    static F2 conj(const F2& a) { return F2{a.c0, zl::neg(a.c1)}; }
    // q-power Frobenius of the untwisted point expressed on the twist (BN tail): (conj(x) xi^((q-1)/3), conj(y) xi^((q-1)/2))
    struct FrobConsts { F2 g2, g3; };
    static FrobConsts make_frob_consts() {
        FrobConsts k;
        const F2 xi{small(PP::ISHIFT), F::one()};  // w^6 = i + ISHIFT
        uint32_t e[FqP::N];
        for (int d : {3, 2}) {  // (q - 1) / 3 and (q - 1) / 2 by long division on 32-bit words
            uint64_t rem = 0;
            uint32_t qm1[FqP::N];
            for (int i = 0; i < FqP::N; i++) qm1[i] = FqP::mod(i);
            qm1[0] -= 1;  // q is odd: no borrow
            for (int i = FqP::N - 1; i >= 0; i--) {
                uint64_t cur = (rem << 32) | qm1[i];
                e[i] = (uint32_t)(cur / d);
                rem = cur % d;
            }
            (d == 3 ? k.g2 : k.g3) = f2_pow(xi, e, 32 * FqP::N);
        }
        return k;
    }
    static G2Aff frob(const G2Aff& Q) {
        static const FrobConsts k = make_frob_consts();  // thread-safe one-time initialisation (C++11 magic static)
        return G2Aff{zl::mul(conj(Q.x), k.g2), zl::mul(conj(Q.y), k.g3), Q.inf};
    }
    // Miller loop value f_{loop,Q}(P) (times subfield factors); P, Q affine Montgomery, neither at infinity
    static Fq12 miller(const F& xP, const F& yP, const G2Aff& Q) {
        Fq12 f = one();
        G2Aff R = Q;
        for (int i = PP::LOOP_BITS - 2; i >= 0; i--) {
            f = mul(sqr(f), line_and_add(R, R, xP, yP));
            if (PP::loop_bit(i)) f = mul(f, line_and_add(R, Q, xP, yP));
        }
        if (PP::BN_TAIL) {
            const G2Aff Q1 = frob(Q);
            G2Aff nQ2 = frob(Q1);
            nQ2.y = zl::neg(nQ2.y);
            f = mul(f, line_and_add(R, Q1, xP, yP));
            f = mul(f, line_and_add(R, nQ2, xP, yP));
        }
        return f;
    }
    // the q^6-power Frobenius: the defining polynomial is a polynomial in w^6, so w -> -w is the automorphism of order two
    static Fq12 conj6(const Fq12& a) {
        Fq12 r = a;
        for (int k = 1; k < 12; k += 2) r.c[k] = zl::neg(a.c[k]);
        return r;
    }
    // a^-1 by linear algebra over Fq: column j of M is a w^j, solve M g = e_0 (Gauss-Jordan, one Fq inversion per pivot: ~2000 products, once per verification)
    // `singular` (optional) reports a == 0: the result is then one() and meaningless (a Miller value is never zero for points of the prime-order subgroups; a caller
    // that was handed garbage learns it here instead of comparing a made-up value -- ADVICE r4)
    static Fq12 inverse(const Fq12& a, bool* singular = nullptr) {
        if (singular) *singular = false;
        F M[12][13];
        for (int j = 0; j < 12; j++) {
            Fq12 b;
            for (auto& x : b.c) x = F::zero();
            b.c[j] = F::one();
            const Fq12 col = mul(a, b);
            for (int i = 0; i < 12; i++) M[i][j] = col.c[i];
        }
        for (int i = 0; i < 12; i++) M[i][12] = i == 0 ? F::one() : F::zero();
        for (int c = 0; c < 12; c++) {
            int piv = c;
            while (piv < 12 && M[piv][c].is_zero()) piv++;
            if (piv == 12) { if (singular) *singular = true; return one(); }  // a is a zero divisor (Fq12 is a field: a == 0)
            if (piv != c)
                for (int k = 0; k < 13; k++) std::swap(M[piv][k], M[c][k]);
            const F inv = zl::inv(M[c][c]);
            for (int k = c; k < 13; k++) M[c][k] = zl::mul(M[c][k], inv);
            for (int i = 0; i < 12; i++) {
                if (i == c || M[i][c].is_zero()) continue;
                const F f = M[i][c];
                for (int k = c; k < 13; k++) M[i][k] = zl::sub(M[i][k], zl::mul(f, M[c][k]));
            }
        }
        Fq12 g;
        for (int i = 0; i < 12; i++) g.c[i] = M[i][12];
        return g;
    }
    // f^((q^12 - 1) / r) = (conj6(f) / f)^((q^6 + 1) / r): the easy factor by the Frobenius and one inversion, the rest as one power in 4-bit fixed windows
    // (~2050 squarings + ~500 products for BLS12-381; rounds 2-3 took the whole 4314-bit power bit by bit with general products: 60 ms per verification)
    static Fq12 final_exp(const Fq12& f0, bool* singular = nullptr) {
        const Fq12 f = mul(conj6(f0), inverse(f0, singular));
        const uint32_t* e = PP::final_exp();
        Fq12 tab[16];
        tab[0] = one();
        tab[1] = f;
        for (int k = 2; k < 16; k++) tab[k] = (k & 1) ? mul(tab[k - 1], f) : sqr(tab[k >> 1]);
        Fq12 acc = one();
        bool started = false;
        for (int i = PP::FINAL_EXP_WORDS * 8 - 1; i >= 0; i--) {
            const uint32_t nib = (e[i >> 3] >> (4 * (i & 7))) & 15u;
            if (started) acc = sqr(sqr(sqr(sqr(acc))));
            if (nib) {
                acc = started ? mul(acc, tab[nib]) : tab[nib];
                started = true;
            }
        }
        return acc;
    }
    // canonical affine words (x||y, and x.c0||x.c1||y.c0||y.c1) -> internal points; all-zero = infinity
    static bool load_g1(const uint64_t* xy, F& x, F& y) {
        memcpy(&x, xy, sizeof(F));
        memcpy(&y, reinterpret_cast<const unsigned char*>(xy) + sizeof(F), sizeof(F));
        if (x.is_zero() && y.is_zero()) return false;
        x = zl::to_mont(x);
        y = zl::to_mont(y);
        return true;
    }
    static bool load_g2(const uint64_t* xy, G2Aff& q) {
        const unsigned char* p = reinterpret_cast<const unsigned char*>(xy);
        memcpy(&q.x.c0, p, sizeof(F));
        memcpy(&q.x.c1, p + sizeof(F), sizeof(F));
        memcpy(&q.y.c0, p + 2 * sizeof(F), sizeof(F));
        memcpy(&q.y.c1, p + 3 * sizeof(F), sizeof(F));
        q.inf = q.x.is_zero() && q.y.is_zero();
        if (q.inf) return false;
        q.x = zl::to_mont(q.x);
        q.y = zl::to_mont(q.y);
        return true;
    }
    // ---- several Miller loops in lock step: every pairing walks the same loop schedule, so the slope denominators of one step (one Fq2 inversion each in
    // miller(): ~270 Fermat inversions per Groth16 verification, a third of its time) share ONE Fq inversion (Montgomery's trick on the Fq2 norms)
    struct Lane { F xP, yP; G2Aff Q, R; };
    // the slope denominator of the step R + S (R == S: tangent); false for a vertical line (cannot happen for points of prime order: the caller falls back)
    static bool slope_den(const G2Aff& R, const G2Aff& S, bool tangent, F2& den) {
        den = tangent ? zl::dbl(R.y) : zl::sub(S.x, R.x);
        return !den.is_zero();
    }
    // as line_and_add for a non-vertical step, with the inverse of its slope denominator supplied
    static Fq12 line_and_add_inv(G2Aff& R, const G2Aff& S, bool tangent, const F2& den_inv, const F& xP, const F& yP) {
        Fq12 l;
        for (auto& x : l.c) x = F::zero();
        F2 m;
        if (tangent) {
            const F2 xx = zl::sqr(R.x);
            m = zl::mul(zl::add(zl::dbl(xx), xx), den_inv);
        } else {
            m = zl::mul(zl::sub(S.y, R.y), den_inv);
        }
        const F2 mx = f2_scale(m, xP);
        const F2 c0 = zl::sub(R.y, zl::mul(m, R.x));
        if (PP::TWIST_DIV) {
            put_fq2(l, 0, c0);
            put_fq2(l, 2, mx);
            l.c[3] = zl::sub(l.c[3], yP);
        } else {
            l.c[0] = zl::neg(yP);
            put_fq2(l, 1, mx);
            put_fq2(l, 3, c0);
        }
        const F2 x3 = zl::sub(zl::sub(zl::sqr(m), R.x), S.x);
        const F2 y3 = zl::sub(zl::mul(m, zl::sub(R.x, x3)), R.y);
        R.x = x3;
        R.y = y3;
        return l;
    }
    // one step of every lane (S_i = R_i for a tangent step, else the given points); false if some line is vertical
    static bool step_all(std::vector<Lane>& L, const std::vector<G2Aff>* S, Fq12& f) {
        const size_t n = L.size();
        std::vector<F2> den(n);
        std::vector<F> norm(n), pre(n);
        for (size_t i = 0; i < n; i++) {
            if (!slope_den(L[i].R, S ? (*S)[i] : L[i].R, S == nullptr, den[i])) return false;
            norm[i] = zl::add(zl::sqr(den[i].c0), zl::sqr(den[i].c1));  // u^2 = -1: (a + b u)^-1 = (a - b u) / (a^2 + b^2)
            pre[i] = i ? zl::mul(pre[i - 1], norm[i]) : norm[i];
        }
        F acc = zl::inv(pre[n - 1]);
        for (size_t k = n; k-- > 0;) {
            const F ninv = k ? zl::mul(acc, pre[k - 1]) : acc;
            if (k) acc = zl::mul(acc, norm[k]);
            const F2 dinv = f2_scale(conj(den[k]), ninv);
            f = mul(f, line_and_add_inv(L[k].R, S ? (*S)[k] : L[k].R, S == nullptr, dinv, L[k].xP, L[k].yP));
        }
        return true;
    }
    static bool miller_multi(std::vector<Lane>& L, Fq12& f) {
        f = one();
        std::vector<G2Aff> Qs(L.size());
        for (size_t i = 0; i < L.size(); i++) { L[i].R = L[i].Q; Qs[i] = L[i].Q; }
        for (int i = PP::LOOP_BITS - 2; i >= 0; i--) {
            f = sqr(f);
            if (!step_all(L, nullptr, f)) return false;
            if (PP::loop_bit(i) && !step_all(L, &Qs, f)) return false;
        }
        if (PP::BN_TAIL) {
            std::vector<G2Aff> Q1(L.size()), nQ2(L.size());
            for (size_t i = 0; i < L.size(); i++) {
                Q1[i] = frob(L[i].Q);
                nQ2[i] = frob(Q1[i]);
                nQ2[i].y = zl::neg(nQ2[i].y);
            }
            if (!step_all(L, &Q1, f) || !step_all(L, &nQ2, f)) return false;
        }
        return true;
    }
    // product of pairings prod_i e(P_i, Q_i) with ONE final exponentiation
    // `degenerate` (optional): set when the Miller value came out as zero (only possible for inputs off the curve / outside the subgroups, e.g. a line through
    // P that vanishes at it); the returned value is then not a pairing
    static Fq12 multi_pairing(const std::vector<const uint64_t*>& ps, const std::vector<const uint64_t*>& qs, bool* degenerate = nullptr) {
        if (degenerate) *degenerate = false;
        std::vector<Lane> L;
        for (size_t i = 0; i < ps.size(); i++) {
            Lane ln;
            if (!load_g1(ps[i], ln.xP, ln.yP) || !load_g2(qs[i], ln.Q)) continue;  // e(O, Q) = e(P, O) = 1
            L.push_back(ln);
        }
        if (L.empty()) return final_exp(one());
        Fq12 f;
        if (!miller_multi(L, f)) {  // a vertical line somewhere (inputs outside the prime-order subgroups): the one-by-one loops handle it
            f = one();
            for (auto& ln : L) f = mul(f, miller(ln.xP, ln.yP, ln.Q));
        }
        return final_exp(f, degenerate);
    }
    static void store(uint64_t* out, const Fq12& a) {  // 12 canonical coefficients
        for (int i = 0; i < 12; i++) {
            const F c = zl::from_mont(a.c[i]);
            memcpy(reinterpret_cast<unsigned char*>(out) + i * sizeof(F), &c, sizeof(F));
        }
    }
};

using BlsEngine = Engine<BLS12_381_Fq, BLS12_381_Pairing>;
using BnEngine = Engine<BN254_Fq, BN254_Pairing>;

}  // namespace pairing
}  // namespace openzl
