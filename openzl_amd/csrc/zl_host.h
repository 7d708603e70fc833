// zl_host.h -- host-side mirror (C++) of the reference's plugin interface for the Groth16 path, above the C ABI.
//
// The reference is Rust; this image has no Rust toolchain, so the operator/plugin surface that sits on top of the
// accelerated path is restated in C++ with the reference's names and argument meaning:
//   openzl::R1CS<F>            plugins/arkworks/src/constraint/mod.rs:64-108 (compiler wrapping a constraint system;
//                              for_contexts :84-90, for_proofs :94-99, Measure :147-177, allocation :210-338)
//   openzl::FpVar<F>           ark-r1cs-std FpVar as the plugin uses it: linear ops cost no constraint, mul costs one
//   openzl::poseidon::*        openzl-crypto/src/poseidon (lfsr.rs:14-100, round_constants.rs:10-59, mds.rs:84-102,
//                              mod.rs:193-282, hash.rs:93-135) + plugin Spec ops plugins/arkworks/src/poseidon/mod.rs:225-298
//   openzl::Groth16<E>         plugins/arkworks/src/groth16.rs:399-467: context_compiler / proof_compiler / compile /
//                              prove / verify with ProvingContext / VerifyingContext / Proof / Error
// Only what config 5 (Poseidon-hash circuit) needs is built; everything else of the gadget stack is out of scope.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <utility>
#include <vector>
#include <mutex>
#include "zl_ctx.h"
#include "zl_pairing.h"

namespace openzl {

// ---- deterministic rng handed to compile / prove (the reference passes `&mut R: CryptoRng + RngCore`) ----------------
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
};
// uniform field element: fill limbs, mask to MODULUS_BITS, reject >= p (ark UniformRand for Fp; canonical here)
template <class FrP>
Fp<FrP> sample_canonical(SplitMix64& rng) {
    for (;;) {
        Fp<FrP> v;
        for (int i = 0; i < FrP::N; i += 2) {
            uint64_t w = rng.next();
            v.l[i] = (uint32_t)w;
            v.l[i + 1] = (uint32_t)(w >> 32);
        }
        const int top_bits = FrP::BITS - 32 * (FrP::N - 1);
        if (top_bits < 32) v.l[FrP::N - 1] &= (1u << top_bits) - 1;
        bool lt = false;
        for (int i = FrP::N - 1; i >= 0; i--) {
            if (v.l[i] < FrP::mod(i)) { lt = true; break; }
            if (v.l[i] > FrP::mod(i)) break;
        }
        if (lt) return v;
    }
}

// ---- linear combinations / variables ----------------------------------------------------------------------------------
// variable key: bit 31 = witness block, low bits = index inside the block; instance index 0 is the constant ONE
static constexpr uint32_t kWitnessBit = 0x80000000u;
static constexpr uint32_t kOne = 0;

template <class FrP>
struct LinearCombination {
    using F = Fp<FrP>;
    std::vector<std::pair<uint32_t, F>> terms;  // sorted by key, no zero coefficients
    static LinearCombination constant(const F& c) {
        LinearCombination r;
        if (!c.is_zero()) r.terms.push_back({kOne, c});
        return r;
    }
    static LinearCombination variable(uint32_t key) {
        LinearCombination r;
        r.terms.push_back({key, F::one()});
        return r;
    }
    bool is_constant() const { return terms.empty() || (terms.size() == 1 && terms[0].first == kOne); }
    LinearCombination add(const LinearCombination& o) const {
        LinearCombination r;
        r.terms.reserve(terms.size() + o.terms.size());
        size_t i = 0, j = 0;
        while (i < terms.size() || j < o.terms.size()) {
            if (j == o.terms.size() || (i < terms.size() && terms[i].first < o.terms[j].first)) r.terms.push_back(terms[i++]);
            else if (i == terms.size() || o.terms[j].first < terms[i].first) r.terms.push_back(o.terms[j++]);
            else {
                F s = zl::add(terms[i].second, o.terms[j].second);
                if (!s.is_zero()) r.terms.push_back({terms[i].first, s});
                i++;
                j++;
            }
        }
        return r;
    }
    LinearCombination scale(const F& c) const {
        LinearCombination r;
        if (c.is_zero()) return r;
        r.terms.reserve(terms.size());
        for (auto& t : terms) r.terms.push_back({t.first, zl::mul(t.second, c)});
        return r;
    }
};

template <class FrP>
struct FpVar {
    using F = Fp<FrP>;
    LinearCombination<FrP> lc;
    F value;  // Montgomery; meaningful in proof mode only
    bool konst = false;  // a constant (no variable in it): tracked beside lc because a witness-only compiler keeps no linear combinations
};

// ---- R1CS<F>: the plugin's compiler (constraint/mod.rs:64-108) -----------------------------------------------------------
template <class FrP>
class R1CS {
public:
    using F = Fp<FrP>;
    using LC = LinearCombination<FrP>;
    enum class Mode { Setup, Prove, ProveWitnessOnly };
    static R1CS for_contexts() { return R1CS(Mode::Setup); }  // SynthesisMode::Setup (constraint/mod.rs:84-90)
    static R1CS for_proofs() { return R1CS(Mode::Prove); }    // SynthesisMode::Prove (constraint/mod.rs:94-99)
    // ark-relations' SynthesisMode::Prove { construct_matrices: false }: the circuit code runs, variables are allocated with their values, but no linear
    // combination is formed and no constraint row stored.  What a prover needs per proof once its ProvingContext holds the circuit's matrices on the device
    // (they are static per circuit): Groth16::prove takes such a compiler for a BOUND context and ships its assignment.  (round 5: 958 465 constraints
    // synthesise in ~0.1 s this way against 1.1 s with the rows; the binding cannot be checked -- no rows to fingerprint -- so a witness-only compiler of another
    // circuit with the same variable counts yields a proof that does not verify, as a wrong witness would.)
    static R1CS for_witness() { return R1CS(Mode::ProveWitnessOnly); }
    bool witness_only() const { return mode_ == Mode::ProveWitnessOnly; }

    // allocation: Public -> instance variable (new_input), Secret -> witness (new_witness) (constraint/mod.rs:302-338)
    FpVar<FrP> new_input(const F& value_mont) {
        instance_.push_back(value_mont);
        if (witness_only()) return FpVar<FrP>{LC{}, value_mont, false};
        return FpVar<FrP>{LC::variable((uint32_t)instance_.size() - 1), value_mont, false};
    }
    FpVar<FrP> new_witness(const F& value_mont) {
        witness_.push_back(value_mont);
        if (witness_only()) return FpVar<FrP>{LC{}, value_mont, false};
        return FpVar<FrP>{LC::variable(kWitnessBit | ((uint32_t)witness_.size() - 1)), value_mont, false};
    }
    FpVar<FrP> constant(const F& c_mont) const { return FpVar<FrP>{witness_only() ? LC{} : LC::constant(c_mont), c_mont, true}; }
    // linear operations: no constraints (plugins/arkworks/src/poseidon/mod.rs:225-274)
    FpVar<FrP> add(const FpVar<FrP>& a, const FpVar<FrP>& b) const {
        if (witness_only()) return FpVar<FrP>{LC{}, zl::add(a.value, b.value), a.konst && b.konst};
        return FpVar<FrP>{a.lc.add(b.lc), zl::add(a.value, b.value), a.konst && b.konst};
    }
    FpVar<FrP> add_const(const FpVar<FrP>& a, const F& c) const {
        if (witness_only()) return FpVar<FrP>{LC{}, zl::add(a.value, c), a.konst};
        return FpVar<FrP>{a.lc.add(LC::constant(c)), zl::add(a.value, c), a.konst};
    }
    FpVar<FrP> mul_const(const FpVar<FrP>& a, const F& c) const {
        if (witness_only()) return FpVar<FrP>{LC{}, zl::mul(a.value, c), a.konst};
        return FpVar<FrP>{a.lc.scale(c), zl::mul(a.value, c), a.konst};
    }
    // multiplication: one constraint a * b = out with a fresh witness; constants fold
    FpVar<FrP> mul(const FpVar<FrP>& a, const FpVar<FrP>& b) {
        // Constant folding: the full compiler decides from the linear combination (no variable term left), the witness-only compiler -- it forms no
        // linear combinations -- from the `konst` flag the operations above propagate.  The two rules differ for a combination whose variable terms CANCEL
        // (x - x: constant for the full compiler, konst = false); a circuit that contains one would allocate witnesses differently in the two modes and a
        // witness-only assignment would not line up with the resident matrices.  The full compiler therefore records any disagreement, and a
        // proving context bound from such a compiler refuses witness-only compilers (Groth16::prove).
        if (!witness_only() && (a.lc.is_constant() != a.konst || b.lc.is_constant() != b.konst)) fold_rules_disagree_ = true;
        if (witness_only() ? a.konst : a.lc.is_constant()) return mul_const(b, a.value);
        if (witness_only() ? b.konst : b.lc.is_constant()) return mul_const(a, b.value);
        FpVar<FrP> out = new_witness(zl::mul(a.value, b.value));
        if (witness_only()) return out;
        A_.push_back(a.lc);
        B_.push_back(b.lc);
        C_.push_back(out.lc);
        return out;
    }
    void enforce_equal(const FpVar<FrP>& a, const FpVar<FrP>& b) {  // a * 1 = b
        if (witness_only()) { equal_ok_ = equal_ok_ && a.value == b.value; return; }
        A_.push_back(a.lc);
        B_.push_back(LC::constant(F::one()));
        C_.push_back(b.lc);
    }
    bool equalities_hold() const { return equal_ok_; }  // witness-only: every enforce_equal saw equal values
    // full compiler: every multiplication folded constants the way a witness-only run of the same circuit code will (see mul)
    bool witness_only_compatible() const { return !fold_rules_disagree_; }
    // Measure (openzl-crypto/src/constraint.rs:151-188; plugin impl constraint/mod.rs:169-177)
    size_t constraint_count() const { return A_.size(); }
    size_t public_variable_count() const { return instance_.size() - 1; }
    size_t secret_variable_count() const { return witness_.size(); }
    size_t num_instance_variables() const { return instance_.size(); }  // ark: includes the constant ONE
    Mode mode() const { return mode_; }

    F eval(const LC& lc) const {
        F acc = F::zero();
        for (auto& t : lc.terms) {
            const F& v = (t.first & kWitnessBit) ? witness_[t.first & ~kWitnessBit] : instance_[t.first];
            acc = zl::add(acc, zl::mul(t.second, v));
        }
        return acc;
    }
    bool is_satisfied() const {
        if (witness_only()) return equal_ok_;  // no rows: only the enforced equalities can be checked
        for (size_t i = 0; i < A_.size(); i++)
            if (zl::mul(eval(A_[i]), eval(B_[i])) != eval(C_[i])) return false;
        return true;
    }
    // Repeated sub-circuits (a hash chain, a Merkle path, ...): append `copies` copies of the constraint rows [r0, r1); in copy j = 1..copies
    // every witness index >= w_from is shifted by j * shift, all other variables (instance block, earlier witnesses) stay.  The caller
    // supplies the values of the copies' witnesses (copies * shift of them, in allocation order).  This is exactly what synthesising the
    // sub-circuit `copies` more times would append when it is wired the same way each time -- without redoing the symbolic
    // linear-combination arithmetic (tests/test_host_mirror.py compares the result with the sequential Python builder).
    void replicate_rows(size_t r0, size_t r1, size_t copies, uint32_t w_from, uint32_t shift, const std::vector<F>& new_witness_values);
    static LC shifted(const LC& lc, uint32_t w_from, uint32_t delta) {
        LC r = lc;
        for (auto& t : r.terms)
            if ((t.first & kWitnessBit) && (t.first & ~kWitnessBit) >= w_from) t.first += delta;  // order of the sorted keys is preserved
        return r;
    }
    uint32_t var_index(uint32_t key) const { return (key & kWitnessBit) ? (uint32_t)instance_.size() + (key & ~kWitnessBit) : key; }
    const std::vector<LC>& rows(int m) const { return m == 0 ? A_ : m == 1 ? B_ : C_; }
    // 64-bit fingerprint of the constraint MATRICES (not the assignment): row / variable counts and the keys + coefficients of EVERY
    // row of every matrix (a sampled digest would let a circuit that differs only in unsampled rows through; hashing the 958 465-row
    // Poseidon chain costs a few ms, once per compiler).  A proving context records it when it learns its circuit and refuses a compiler whose
    // fingerprint differs: two different circuits of the same shape must not silently share device matrices.  Rows are append-only, so
    // the value is cached per (row count, variable counts): prove() pays for it once per compiler.
    uint64_t structure_digest() const {
        static std::mutex digest_mu;  // one compiler may be proven from several lanes at once (Groth16::prove(..., lane)): the cache below is filled once, under this lock
        std::lock_guard<std::mutex> lk(digest_mu);
        if (digest_rows_ == A_.size() && digest_inst_ == instance_.size() && digest_wit_ == witness_.size()) return digest_;
        uint64_t h = 0xCBF29CE484222325ull;
        auto mix = [&h](uint64_t v) { h = (h ^ v) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; };
        mix(A_.size()); mix(instance_.size()); mix(witness_.size());
        const size_t rows = A_.size();
        for (int m = 0; m < 3; m++) {
            const std::vector<LC>& M = this->rows(m);
            for (size_t i = 0; i < rows; i++) {
                mix(M[i].terms.size());
                for (const auto& t : M[i].terms) { mix(t.first); for (int k = 0; k < F::N; k += 2) mix((uint64_t)t.second.l[k] | ((uint64_t)t.second.l[k + 1] << 32)); }
            }
        }
        digest_ = h;
        digest_rows_ = A_.size(); digest_inst_ = instance_.size(); digest_wit_ = witness_.size();
        return h;
    }
    // TEST ONLY (zl_test_circuit_tweak): doubles the first coefficient of row 0 of A -- a different circuit of the same shape
    void tweak_for_tests() {
        if (!A_.empty() && !A_[0].terms.empty()) A_[0].terms[0].second = zl::add(A_[0].terms[0].second, A_[0].terms[0].second);
        digest_rows_ = (size_t)-1;
    }
    const std::vector<F>& instance_assignment() const { return instance_; }
    const std::vector<F>& witness_assignment() const { return witness_; }

private:
    explicit R1CS(Mode m) : mode_(m) { instance_.push_back(F::one()); }
    Mode mode_;
    bool equal_ok_ = true, fold_rules_disagree_ = false;
    std::vector<F> instance_, witness_;
    std::vector<LC> A_, B_, C_;
    mutable uint64_t digest_ = 0;
    mutable size_t digest_rows_ = (size_t)-1, digest_inst_ = 0, digest_wit_ = 0;
};

// ---- Poseidon (config 5) ------------------------------------------------------------------------------------------------------
namespace poseidon {
// 80-bit Grain LFSR in self-shrinking mode (openzl-crypto/src/poseidon/lfsr.rs:14-100)
class GrainLFSR {
public:
    GrainLFSR(unsigned modulus_bits, unsigned width, unsigned rf, unsigned rp) {
        for (auto& b : state_) b = false;
        head_ = 0;
        append(2, 1);
        append(4, 0);
        append(12, modulus_bits);
        append(12, width);
        append(10, rf);
        append(10, rp);
        append(30, (1u << 30) - 1);
        for (int i = 0; i < 160; i++) update();
    }
    bool next() {
        bool bit = update();
        while (!bit) {
            update();
            bit = update();
        }
        return update();
    }

private:
    bool state_[80];
    unsigned head_;
    void append(unsigned n, uint64_t bits) {
        for (int i = (int)n - 1; i >= 0; i--) set_next((bits >> i) & 1);
    }
    bool set_next(bool b) {
        state_[head_] = b;
        head_ = (head_ + 1) % 80;
        return b;
    }
    bool bit(unsigned i) const { return state_[(i + head_) % 80]; }
    bool update() { return set_next(bit(62) ^ bit(51) ^ bit(38) ^ bit(23) ^ bit(13) ^ bit(0)); }
};

template <class FrP>
struct Constants {
    using F = Fp<FrP>;
    static constexpr int WIDTH = 3, FULL_ROUNDS = 8, PARTIAL_ROUNDS = 55;  // arity 2 (plugins/arkworks/src/poseidon/mod.rs:300-304)
    std::vector<F> round_keys;  // WIDTH * (RF + RP), Montgomery
    F mds[WIDTH][WIDTH];
    Constants() {
        // generate_round_constants: MODULUS_BITS bits big-endian per candidate, rejection (round_constants.rs:10-59)
        GrainLFSR lfsr(FrP::BITS, WIDTH, FULL_ROUNDS, PARTIAL_ROUNDS);
        while ((int)round_keys.size() < WIDTH * (FULL_ROUNDS + PARTIAL_ROUNDS)) {
            F v = F::zero();
            for (int b = FrP::BITS - 1; b >= 0; b--)
                if (lfsr.next()) v.l[b >> 5] |= 1u << (b & 31);
            bool lt = false;
            for (int i = FrP::N - 1; i >= 0; i--) {
                if (v.l[i] < FrP::mod(i)) { lt = true; break; }
                if (v.l[i] > FrP::mod(i)) break;
            }
            if (lt) round_keys.push_back(zl::to_mont(v));
        }
        // generate_mds: M[i][j] = 1 / (i + (t + j)) (mds.rs:84-102)
        for (int i = 0; i < WIDTH; i++)
            for (int j = 0; j < WIDTH; j++) mds[i][j] = zl::inv(zl::from_u64<FrP>((uint64_t)(i + WIDTH + j)));
    }
};
// native permutation (openzl-tutorials/src/poseidon.rs:165-222 schedule; COM = ())
template <class FrP>
void permute_native(const Constants<FrP>& c, Fp<FrP> state[3]) {
    using F = Fp<FrP>;
    const int half = c.FULL_ROUNDS / 2;
    for (int rnd = 0; rnd < c.FULL_ROUNDS + c.PARTIAL_ROUNDS; rnd++) {
        for (int i = 0; i < 3; i++) state[i] = zl::add(state[i], c.round_keys[3 * rnd + i]);
        const int lanes = (rnd < half || rnd >= half + c.PARTIAL_ROUNDS) ? 3 : 1;
        for (int i = 0; i < lanes; i++) {
            F x2 = zl::sqr(state[i]), x4 = zl::sqr(x2);
            state[i] = zl::mul(x4, state[i]);
        }
        F nx[3];
        for (int i = 0; i < 3; i++) {
            F acc = zl::mul(c.mds[i][0], state[0]);
            acc = zl::add(acc, zl::mul(c.mds[i][1], state[1]));
            nx[i] = zl::add(acc, zl::mul(c.mds[i][2], state[2]));
        }
        for (int i = 0; i < 3; i++) state[i] = nx[i];
    }
}
// the same permutation, recording what the in-circuit version allocates: x^2, x^4, x^5 of every S-box whose input is not a constant
// (lane 0 of the first round is: the capacity element 2^arity - 1 plus a round key), in allocation order
template <class FrP>
void permute_native_record(const Constants<FrP>& c, Fp<FrP> state[3], std::vector<Fp<FrP>>& rec) {
    using F = Fp<FrP>;
    const int half = c.FULL_ROUNDS / 2;
    for (int rnd = 0; rnd < c.FULL_ROUNDS + c.PARTIAL_ROUNDS; rnd++) {
        for (int i = 0; i < 3; i++) state[i] = zl::add(state[i], c.round_keys[3 * rnd + i]);
        const int lanes = (rnd < half || rnd >= half + c.PARTIAL_ROUNDS) ? 3 : 1;
        for (int i = 0; i < lanes; i++) {
            const F x2 = zl::sqr(state[i]), x4 = zl::sqr(x2);
            state[i] = zl::mul(x4, state[i]);
            if (rnd != 0 || i != 0) { rec.push_back(x2); rec.push_back(x4); rec.push_back(state[i]); }
        }
        F nx[3];
        for (int i = 0; i < 3; i++) {
            F acc = zl::mul(c.mds[i][0], state[0]);
            acc = zl::add(acc, zl::mul(c.mds[i][1], state[1]));
            nx[i] = zl::add(acc, zl::mul(c.mds[i][2], state[2]));
        }
        for (int i = 0; i < 3; i++) state[i] = nx[i];
    }
}
// in-circuit arity-2 hash (Hasher::hash, hash.rs:123-135): state = (2^arity - 1, x, y), output = lane 0
template <class FrP>
FpVar<FrP> hash(const Constants<FrP>& c, const FpVar<FrP>& x, const FpVar<FrP>& y, R1CS<FrP>& compiler) {
    using V = FpVar<FrP>;
    V state[3] = {compiler.constant(zl::from_u64<FrP>(3)), x, y};
    const int half = c.FULL_ROUNDS / 2;
    for (int rnd = 0; rnd < c.FULL_ROUNDS + c.PARTIAL_ROUNDS; rnd++) {
        for (int i = 0; i < 3; i++) state[i] = compiler.add_const(state[i], c.round_keys[3 * rnd + i]);
        const int lanes = (rnd < half || rnd >= half + c.PARTIAL_ROUNDS) ? 3 : 1;
        for (int i = 0; i < lanes; i++) {  // apply_sbox = x^5 (plugins/arkworks/src/poseidon/mod.rs:287-298)
            V x2 = compiler.mul(state[i], state[i]);
            V x4 = compiler.mul(x2, x2);
            state[i] = compiler.mul(x4, state[i]);
        }
        V nx[3];
        for (int i = 0; i < 3; i++) {
            V acc = compiler.mul_const(state[0], c.mds[i][0]);
            acc = compiler.add(acc, compiler.mul_const(state[1], c.mds[i][1]));
            nx[i] = compiler.add(acc, compiler.mul_const(state[2], c.mds[i][2]));
        }
        for (int i = 0; i < 3; i++) state[i] = nx[i];
    }
    return state[0];
}
}  // namespace poseidon

// ---- pairing-engine configs ---------------------------------------------------------------------------------------------------
struct Bls12_381 {
    using FrP = BLS12_381_Fr;
    using G1 = BlsG1;
    using G2 = BlsG2;
    using PairingP = BLS12_381_Pairing;
    static constexpr zl_curve_t curve = ZL_BLS12_381;
};
struct Bn254 {
    using FrP = BN254_Fr;
    using G1 = BnG1;
    using G2 = BnG2;
    using PairingP = BN254_Pairing;
    static constexpr zl_curve_t curve = ZL_BN254;
};

// ---- Groth16<E>: ProofSystem (groth16.rs:405-467) -------------------------------------------------------------------------------
struct Error {
    int code;  // the reference's Error is an opaque unit struct (groth16.rs:35-45); the code is extra, for diagnostics
};
template <class T>
struct Result {
    bool ok;
    T value;
    Error error;
};

template <class FrP> struct R1csExport;
template <class E>
struct Groth16 {
    using FrP = typename E::FrP;
    using F = Fp<FrP>;
    using Compiler = R1CS<FrP>;
    struct Trapdoor { F alpha, beta, gamma, delta, tau; };  // canonical
    struct ProvingContext {                                 // ProvingContext<E>(pub ProvingKey<E>) groth16.rs:127-140
        zl_ctx* ctx = nullptr;
        uint64_t a_query = 0, b_g1_query = 0, h_query = 0, l_query = 0, b_g2_query = 0;
        // device-resident constraint matrices (static per circuit): uploaded by compile; a context decoded from bytes does not know its
        // circuit, so the first prove() uploads them and BINDS the context to that circuit (hence mutable: the first prove of a decoded
        // context mutates it, so it must not race with another prove on the same context; every later prove is read-only).
        // circuit_digest = R1CS::structure_digest() of the bound circuit: a compiler with another fingerprint is refused.
        mutable uint64_t r1cs = 0;
        mutable size_t n_constraints = 0;
        mutable uint64_t circuit_digest = 0;
        mutable bool witness_only_ok = true;  // the bound circuit folds constants alike in both compiler modes (R1CS::witness_only_compatible)
        std::vector<uint64_t> alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2;
        size_t n_instance = 0, n_witness = 0, domain_size = 0;
        Trapdoor trapdoor;  // kept ONLY so tests can recompute proofs in the exponent; a real setup drops it
        bool has_trapdoor = false;  // false for a context decoded from bytes
    };
    struct VerifyingContext {  // ark_groth16::VerifyingKey (the reference holds it prepared, groth16.rs:181-186); canonical affine points
        std::vector<uint64_t> alpha_g1, beta_g2, gamma_g2, delta_g2;
        std::vector<uint64_t> gamma_abc_g1;  // n_instance points, index 0 pairs with the constant ONE
        std::vector<F> gamma_abc_exponents;  // discrete logs, kept for exponent checks in tests
    };
    using Input = std::vector<F>;
    using Proof = zl_g16_proof;

    static Compiler context_compiler() { return Compiler::for_contexts(); }  // groth16.rs:418-420
    static Compiler proof_compiler() { return Compiler::for_proofs(); }      // groth16.rs:423-425
    // `exported`: the CSR export of `compiler` when the caller already holds one (the C hooks do): saves rebuilding it for the matrix upload
    static Result<std::pair<ProvingContext, VerifyingContext>> compile(zl_ctx* ctx, const Compiler& compiler, SplitMix64& rng,
                                                                        const R1csExport<FrP>* exported = nullptr);
    // lane: the ctx the proof runs on -- the context's own (default) or a fork of it (zl_ctx_fork): N host threads, one lane each, prove side by side over
    // ONE device-resident key, as N threads may share the reference's `&ProvingContext`.  A context decoded from bytes is bound (first proof) on its own ctx.
    static Result<Proof> prove(const ProvingContext& context, const Compiler& compiler, SplitMix64& rng, F* r_out = nullptr, F* s_out = nullptr,
                               zl_ctx* lane = nullptr);
    // verify (groth16.rs:459-466): e(A, B) == e(alpha, beta) e(sum_i x_i gamma_abc_i, gamma) e(C, delta); input = public inputs (canonical)
    static Result<bool> verify(const VerifyingContext& vk, const Input& input, const Proof& proof);
    static void release(ProvingContext& context);
    // Wire format of ProvingContext<E> (codec::Encode / Decode, groth16.rs:142-179): ark_groth16::ProvingKey<E> written with
    // serialize_unchecked = the uncompressed form (zl_serialize.h).  The key carries its VerifyingKey, so encode takes both halves of
    // compile()'s result and decode returns both.  decode uploads the five queries (flags: ZL_CHECK verifies every point on the
    // device) and builds the same window tables as compile; like the reference's deserialize_unchecked it trusts the bytes otherwise.
    static Result<std::vector<uint8_t>> encode(const ProvingContext& context, const VerifyingContext& vk);
    static Result<std::pair<ProvingContext, VerifyingContext>> decode(zl_ctx* ctx, const uint8_t* bytes, size_t len, unsigned flags = 0);
    // ark_groth16::VerifyingKey<E>::serialize (compressed): alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1
    static std::vector<uint8_t> encode_verifying_key(const VerifyingContext& vk);
    static int build_window_tables(const ProvingContext& context);  // the per-query tables of a large key (compile and decode)
};

// CSR export of a compiler (what ark hands to the prover as ConstraintMatrices + assignments)
template <class FrP>
struct R1csExport {
    std::vector<uint32_t> ptr[3], col[3];
    std::vector<uint64_t> val[3];
    std::vector<uint64_t> assignment;
    zl_r1cs view{};
    void build(const R1CS<FrP>& cs);
};

}  // namespace openzl
