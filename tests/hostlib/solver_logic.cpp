// Host build (g++) of the DEVICE executor's instruction semantics (csrc/solver_instr.cuh) so that the CPU suite checks them against the
// builder's Python-integer wire values and against the host executor without a GPU: the levels are walked serially (optionally every
// level back to front: the instructions of a level must not depend on each other).  Test code only.
#include "solver_instr.cuh"
#include "../../zkmerkle-proof-of-solvency_amd/host/r1cs_file.hpp"
#include "../../zkmerkle-proof-of-solvency_amd/host/solver_file.hpp"
#include <string.h>
#include <string>
#include <vector>
using namespace zk;
using namespace zkpor_host;

extern "C" void sl_fr_inverse(const Fr* a, Fr* o, size_t n) { for (size_t i = 0; i < n; ++i) o[i] = fr_inverse(a[i]); }
extern "C" void sl_fr_inv_fermat(const Fr* a, Fr* o, size_t n) { for (size_t i = 0; i < n; ++i) o[i] = Fr::inv(a[i]); }

extern "C" int sl_run(const uint8_t* r1cs, size_t r1cs_len, const uint8_t* solv, size_t solv_len, const uint64_t* inputs, size_t n_inputs,
                      const uint32_t* pre_ids, const uint64_t* pre_vals, size_t n_pre, int back_to_front, uint64_t* w_out, uint64_t info[2]) {
    R1csFileView r; SolverView s; std::string why;
    if (ParseR1csFile(r1cs, r1cs_len, &r, &why) != 0 || ParseSolverFile(solv, solv_len, &s, &why) != 0) return 1;
    if (n_inputs != r.n_public + r.n_secret) return 1;
    std::vector<uint8_t> ckind(r.n_coeff, 0), hk(s.hint_names.size(), 0), known(r.n_wires, 0);
    const Fr* tab = (const Fr*)r.coeff;
    const Fr one = Fr::one(), mone = Fr::neg(one);
    for (size_t i = 0; i < r.n_coeff; ++i) ckind[i] = tab[i].is_zero() ? 3 : (tab[i] == one ? 1 : (tab[i] == mone ? 2 : 0));
    for (size_t i = 0; i < hk.size(); ++i) hk[i] = hint_kind_of_name(s.hint_names[i].c_str());
    SolverProg P;
    P.coeff = tab; P.ckind = ckind.data();
    for (int m = 0; m < 3; ++m) { P.row_ptr[m] = r.row_ptr[m]; P.cid[m] = r.coeff_ids[m]; P.wid[m] = r.wire_ids[m]; }
    P.n_constraints = (u32)r.n_constraints; P.n_wires = (u32)r.n_wires; P.n_coeff = (u32)r.n_coeff;
    std::vector<uint32_t> kinds(s.n_instructions);
    for (uint64_t i = 0; i < s.n_instructions; ++i) kinds[i] = InstrKind(s, i);      // what csrc/solver.hip uploads
    P.kind = kinds.data(); P.arg = s.arg; P.calldata = s.calldata; P.n_calldata = s.n_calldata;
    P.hint_kind = hk.data(); P.n_hint_names = (u32)hk.size();
    Fr* w = (Fr*)w_out;
    memset(w_out, 0, r.n_wires * 32);
    memcpy(w_out, inputs, n_inputs * 32);
    for (size_t i = 0; i < n_inputs; ++i) known[i] = 1;
    for (size_t i = 0; i < n_pre; ++i) { if (pre_ids[i] >= r.n_wires) return 1; memcpy(&w[pre_ids[i]], pre_vals + 4 * i, 32); known[pre_ids[i]] = 1; }
    for (uint64_t l = 0; l < s.n_levels; ++l) {
        const uint64_t lo = s.level_ptr[l], hi = s.level_ptr[l + 1];
        for (uint64_t k = 0; k < hi - lo; ++k) {
            const uint32_t ins = s.level_instr[back_to_front ? hi - 1 - k : lo + k];
            const int rc = solve_instr(P, ins, w, known.data());
            if (rc) { info[0] = ins; info[1] = l; return rc; }
        }
    }
    for (size_t i = 0; i < r.n_wires; ++i) if (!known[i]) { info[0] = i; return 30; }
    return 0;
}
