/* zkpor.h — C ABI of the MI355X (gfx950) backend for the reference's Groth16 / Poseidon hot path.
 *
 * Drop-in boundary.  Every entry point replaces work the reference does inside ONE call:
 *   groth16.Prove(r1cs, pk, witness)            src/prover/prover/prover.go:269   -> zkpor_prove_tail (+ zkpor_commit)
 *   pk.UnsafeReadFrom / LoadSnarkParamsOnce     src/prover/prover/prover.go:285-367 -> zkpor_pk_* (one-time HBM upload)
 *   pk.WriteTo / UnsafeReadFrom container         src/keygen/main.go:46, prover.go:343 -> zkpor_pk_load_gnark(_mem), zkpor_pk_gnark_layout
 *   proof.WriteRawTo                            src/prover/prover/prover.go:201   -> zkpor_proof_write_raw
 *   AccountInfoToHash / buildAccountTree        src/utils/utils.go:744-750, src/witness/main.go:130-199 -> zkpor_poseidon_leaves
 *   FixedDepthMerkleTree.Build / Root           src/utils/merkletree/merkletree.go:192-279 -> zkpor_merkle_build
 *   poseidon.Poseidon / hash.Hash Write+Sum     src/utils/account_tree.go:19,27 (hasher factory) -> zkpor_poseidon_hash
 *   merkletree.FixedDepthMerkleTree (object)    src/utils/merkletree/merkletree.go:137-355, account_tree.go:14-29 -> zkpor_tree_*
 *   account totals, collateral tier claims      src/utils/utils.go:608-615,648-685, circuit/utils.go:227-278 -> zkpor_account_totals
 *   Witness.Run per-batch commitments           src/witness/witness/witness.go:159-198, utils.go:26-88,779-800 -> zkpor_cex_commitments,
 *                                                                                                 zkpor_batch_commitments
 *   compressed key arrays (pk.WriteTo form)     src/keygen/main.go:46, prover.go:336-349 -> zkpor_pk_set_g1/g2_compressed
 *   constraint evaluation a, b, c = L.w, R.w, O.w (inside groth16.Prove, prover.go:269) -> zkpor_r1cs_*
 * The cgo binding a maintainer adds on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - all functions return 0 on success, <0 on error (ZKPOR_E_*); they never throw — every entry point is a function-try-block
 *     (csrc/common.cuh ZK_ABI_CATCH, kept complete by tools/abi_firewall.py --check): std::bad_alloc is ZKPOR_E_OOM, any other C++
 *     exception ZKPOR_E_HIP, its text is in the next zkpor_last_error of the calling thread — never call back into the
 *     host runtime and never retain a host pointer after returning (cgo pointer-passing rule).
 *   - field elements are gnark-crypto's in-memory form: 4 x uint64 little-endian limbs, MONTGOMERY form
 *     (fr.Element / fp.Element).  G1 affine = X,Y (64 B); G2 affine = X.A0,X.A1,Y.A0,Y.A1 (128 B);
 *     G1 Jacobian = X,Y,Z (96 B); G2 Jacobian 192 B.  Affine infinity = all-zero.
 *   - a zkpor_ctx is bound to one GPU and one HIP stream and is single-caller (one call at a time per context); different
 *     contexts are independent — one per GPU, or several per GPU to keep more than one proof in flight.  HIP's current
 *     device is a per-host-thread setting; every entry point that takes a context / key / tree / R1CS handle switches the
 *     calling thread to the handle's GPU for the duration of the call and restores the previous device on return, so a
 *     handle may be used from ANY thread (a goroutine need not be locked to an OS thread) and its allocations, events and
 *     launches always land on its own GPU.  A key belongs to the GPU of the context it was created with; proving with a
 *     context of another GPU is ZKPOR_E_ARG.
 *   - zkpor_host_register / zkpor_dev_upload_async are the only calls that keep reading a host range after they
 *     return (until zkpor_sync): the caller pins that memory for exactly that reason.
 *   - *_dev variants take DEVICE pointers (inputs already resident in HBM) and are asynchronous on the
 *     context's stream unless they return a result to the host.
 *   - there is NO CPU fallback: without a usable gfx950 device every call fails with ZKPOR_E_NODEVICE.
 */
#ifndef ZKPOR_H
#define ZKPOR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKPOR_OK 0
#define ZKPOR_E_NODEVICE (-1)
#define ZKPOR_E_HIP (-2)      /* a HIP runtime call failed; zkpor_last_error has the text */
#define ZKPOR_E_ARG (-3)      /* invalid argument */
#define ZKPOR_E_OOM (-4)      /* device allocation failed (never aborts the process) */
#define ZKPOR_E_STATE (-5)    /* key not fully loaded for the requested operation */

typedef struct zkpor_ctx zkpor_ctx;
typedef struct zkpor_pk zkpor_pk;

/* point arrays of the Groth16 proving key (gnark groth16_bn254.ProvingKey: G1.A, G1.B, G1.K, G1.Z, G2.B and the
 * Pedersen CommitmentKeys Basis / BasisExpSigma) */
enum { ZKPOR_G1_A = 0, ZKPOR_G1_B = 1, ZKPOR_G1_K = 2, ZKPOR_G1_Z = 3, ZKPOR_G1_COMMIT_BASIS = 4,
       ZKPOR_G1_COMMIT_BASIS_SIGMA = 5, ZKPOR_G1_NUM = 6 };
enum { ZKPOR_G2_B = 0, ZKPOR_G2_NUM = 1 };
/* order of pk.G1.Z relative to the coefficient index of h: gnark >= 0.9 stores it bit-reversed at setup */
enum { ZKPOR_Z_ORDER_BITREV = 0, ZKPOR_Z_ORDER_NATURAL = 1 };

/* ---- ABI version -------------------------------------------------------------------------------------- */
/* Bumped whenever an existing entry point changes its signature or meaning (3: z_order joined the zkpor_pk_load_gnark* family in
 * the middle of their argument lists; 4: zkpor_compute_h_shard_dev under the default "ntt_h" 1 — c is not exchanged after step 1 and
 * step 3 takes it).  A binding compiled or written against another value must refuse to run: a stale ctypes /
 * cgo caller would otherwise pass arguments in the old positions and nothing would notice at load time.  zkpor.py and
 * go/zkporgpu check it when the library is loaded. */
#define ZKPOR_ABI_VERSION 4u
uint32_t zkpor_abi_version(void);

/* ---- context ------------------------------------------------------------------------------------------- */
/* stream: a hipStream_t (as void*) the context launches on, or NULL to create its own. */
int32_t zkpor_init(int device, void* stream, zkpor_ctx** out);
void zkpor_destroy(zkpor_ctx* ctx);
const char* zkpor_last_error(zkpor_ctx* ctx);
int32_t zkpor_sync(zkpor_ctx* ctx);
/* hand the context's grow-only scratch back to the device: the workspace of the multi-exponentiations (digit streams, accumulation regions), the
 * staging area of the host-pointer calls, the NTT domains' twiddle tables.  Everything is re-created on demand by the next call that needs it
 * (a 2^26 proof: ~0.1 s once).  Waits for the context's streams first.  Keys, trees, matrices and solver programs are handles of their own and stay.
 * For a prover that switches tiers (the reference loads another key when the batch shape changes, src/prover/prover/prover.go:60-90: the old
 * tier's 40-60 GB of scratch would otherwise stay beside the new key) and for a process that shares its GPU with another one. */
int32_t zkpor_trim(zkpor_ctx* ctx);
/* tuning knobs: "msm_window" (bits, 0 = auto), "msm_chunk" (entries per accumulation thread),
 * "msm_tables" (1..8, default 1; applies to keys loaded AFTER it is set): m > 1 stores every key array as m interleaved
 * fixed-base tables, entry i*m + q = 2^(q * piece * c) P_i, so that the digits of a scalar share ceil(W / m) bucket windows: at
 * 2^26, m = 4 costs 112 GB of HBM instead of 28 and buys 12 digits of 22 bits instead of 13 of 20 at the same number of buckets
 * (6 % fewer bucket additions; built on the device at load time, ~14 s).  Such a key cannot be cut into shards, and "msm_window" must
 * not change between the load and the proofs,
 * "copy_threads" (how PAGEABLE host memory crosses PCIe in the host-pointer entry points: n > 0, default 4 since round 5, copies through
 * the context's pinned bounce buffers with n host threads — 30 GB/s, and the GPU never reads a page the library does not own; 0 hands the
 * range to the HIP runtime, which page-locks it on the fly and lets the DMA engine read the caller's pages — 56 GB/s measured, but pages the
 * kernel may still migrate (a transparent-huge-page collapse under the copy: DESIGN.md §6c) are then read by the GPU.  Memory page-locked
 * ONCE by the caller (zkpor_host_register, the recommended form for a Go prover's long-lived slices) is always handed to the DMA engine
 * directly, whatever this parameter says.  Every other entry point that takes host data — key arrays, matrices, inputs — copies
 * through the bounce buffers as well),
 * "msm_chunk" 0 = automatic (64 from 2^22 scalars up, else 32), otherwise 4..4096 (below 4 the partial-sum recursion does not shrink),
 * "msm_tail_chunk" (8, the default: entries per thread of the partial-sum levels below 2^21 entries — their duration is one thread's
 * chain of additions; 0 = "msm_chunk" everywhere; 4..64),
 * "msm_reduce_scan" (1, the default: the small levels of the bucket reduction run one lane — G2: one lane pair — per bucket; 2: G1
 * only; 0: the serial walk),
 * "msm_filter" (1, the default: B1 / B2 and K accumulate from the witness digit stream minus the entries of their absent points),
 * "msm_chain" (2, the default since round 6: everything of a prove-tail sum after its level-1 kernel — the partial-sum levels, the bucket reduction,
 * the copies of the finals: ~40 short dependent launches — runs on a second stream beside the NEXT sum's level-1 kernel, the sums' workspace in two
 * regions that take turns: 318 -> 298 ms per zkpor50_1380 proof with two workers, 351 -> 298 ms per proof for two host-pointer callers.  The chain stream
 * has a hardware queue of its own: on an ORDINARY stream — the runtime deals those onto four hardware queues — its short launches queued behind long
 * kernels of other streams and the tail got slower (281 -> 359 ms, the first version).  1: only tails on "tail_streams" / "tail_reserve_cus" streams;
 * 0: one stream, one region, every sum behind the one before),
 * "r1cs_order" (1, the default since round 6: a, b, c = L.w, R.w, O.w walk a matrix's rows by shape — term count, then the pattern of coefficient
 * kinds — so that the rows of a wave run the same iterations; 0: natural order.  The results are the same bits),
 * the digit-stream sort's (csrc/sort.hip; rocPRIM's "sort_block" of rounds 3-5 is accepted and ignored): "sort_grid" (0, the default: the library's
 * choice; else the number of its persistent 256-thread workgroups — what the sort costs the main stream's kernels is RESIDENCE),
 * "sort_stage" (1: a tile's entries are staged through LDS and leave as whole runs, 40 KB per workgroup; 0: straight to memory from a 4 KB
 * workgroup), "sort_tile" (0 = 4096: entries staged at a time, 1024 / 2048 / 4096), "sort_generic" (0; 1: the runtime-window level 0 even for the
 * shapes that have a compile-time one — tests compare the two),
 * "ntt_twiddles" (0, the default: the inter-pass twiddles of every index field are read from its table; 1: fields whose table exceeds 16 MiB — the
 * highest field of a 2^26 domain, 2 GiB per direction — generate them from two half tables, one more field product per element and 15 GB less
 * HBM traffic per computeH; measured: no faster; 2: every field, for tests),
 * "ntt_fuse" (1, the default: computeH's neighbouring passes over one index field run as one kernel),
 * "ntt_h" (1, the default since round 6: computeH runs SIX transforms — the inverse coset transform is linear, so c's coefficients, which its
 * inverse transform has produced anyway, are subtracted behind it and c never goes to the coset; 0: gnark's seven.  The same h bit for bit,
 * for every a, b, c),
 * "gpu_token" (1, the default: host-pointer proofs of several contexts on one GPU take turns on the device, see zkpor_prove_tail;
 * 0: their kernels share it freely), "host_order" (0, the default: a proof that finds the GPU free sends w first and a, b, c
 * underneath its own witness sums; 1: always everything first),
 * "poseidon_out_idx", "poseidon_carry_idx" (hash-wrapper convention),
 * the solver executor's (zkpor_solver_*): "solver_poseidon" (1, the default: a Poseidon call runs on 16 lanes; 0: in one thread),
 * "poseidon_defer" (64, the default: a launch of up to this many 16-lane calls — the challenge sponge, the CEX chains — parks the S-box inputs raw and a
 * wide kernel behind it converts them and writes wires and rows; same bits; 0 = the waves convert as they go),
 * "solver_batch_from" (2^21: levels from this many generic instructions on run four per thread), "solver_chain" (1, the default: runs of
 * one-instruction levels are decoded side by side and executed from registers; 0: the narrow-level kernel), "solver_tree_from" (1 024: levels from this many generic instructions on share one field inversion per workgroup), "solver_beside" (1, the default: a Poseidon call that carries a join level runs on a side stream beside the levels up to it; 0: in place), "solver_pre_join" (1, the default since round 6: the input expressions of such a call are evaluated side by side in front of it; 0: inside the serial kernel, one lane at a time), "solver_long" (256: those levels leave constraints of more terms than this to a wave each; 0 = never), "solver_defer_checks"
 * (1, the default: see zkpor_solver_set_abc_dev; 0: a run executes its CHECK instructions even when a, b, c buffers are set),
 * "poseidon_coop" (-1, the default: account leaves and CEX commitments run 16 lanes per hash chain when a launch has fewer than
 * 65 536 chains; 0 never, 1 always),
 * "tail_reserve_cus" (0, the default; a multiple of 8 up to 128: the kernels of the prove tail — NTT passes, digit streams, bucket
 * accumulations — run on HIP streams whose CU mask leaves that many compute units free, R / 8 on each XCD, so that the narrow
 * dependent launches of ANOTHER worker context's solver program start at once instead of queueing behind full-size MSM grids:
 * solve(i + 1) beside tail(i) with two workers per GPU — host/prover_host.hpp, bench.py `end_to_end`: 380 -> 340 ms per zkpor50_1380 proof.  With it
 * a prove tail takes the device turn ("gpu_token": one tail at a time, the other worker's solve beside it) and the long streams get hardware
 * queues of their own.  It applies to the device split only (zkpor_solver_* ... zkpor_prove_tail_dev): a host-pointer call holds the device turn
 * over its own solve AND tail, nothing runs beside it, and its tail keeps every compute unit.  The value may change between proofs: a context
 * keeps one pair of masked streams per value it has had (at most four different non-zero values; a fifth is ZKPOR_E_STATE and leaves the setting
 * as it was) and destroys none of them before zkpor_destroy — round 5's destroy-and-re-create crashed inside the HIP runtime),
 * "tail_streams" (0; 1: the prove tail runs on its own streams — hardware queues of their own — and takes the device turn even WITHOUT a
 * reserve, "tail_reserve_cus" 0: the other worker's solver launches compete for compute units as they free up instead of owning a share),
 * "stream_priority" (0; 1: the context's own stream — solver, a / b / c, commitment — is re-created with the highest stream priority; only
 * for contexts created without a caller's stream; measured: changes nothing next to "tail_streams"),
 * "stream_own_queue" (1, the default since round 6 for contexts created without a caller's stream: the context's own stream — solver levels, a / b / c, the
 * commitment, an unmasked prove tail — has a hardware queue of its own.  Ordinary HIP streams are dealt onto four hardware queues in creation order, so
 * whether two workers' streams share a queue, each waiting behind the other's launches, was an accident of what the process had created before: the
 * same two-worker region 327 or 308 ms per proof, two tails in flight 300 or 272 ms (profiles/r06_stream_own_queue_ab.json).  Such a stream synchronises with
 * the legacy NULL stream: a caller that launches on the NULL stream serialises with it.  0: an ordinary stream, the round-5 behaviour),
 * "tail_digits_early" (1, the default: a tail that finds another worker's tail on the device builds its digit stream of w BEFORE it waits for
 * the turn, beside that tail's accumulations, instead of beside its own NTT passes; 0: after the turn),
 * "debug_validate" (0; 1: every sorted digit stream is checked on the device before its accumulation reads it — keys ascending and
 * below the bucket count, point indices inside the key array — and a violation is ZKPOR_E_STATE instead of a GPU memory fault) */
int32_t zkpor_set_param(zkpor_ctx* ctx, const char* name, int64_t value);
/* per-phase GPU time in ms accumulated since the last reset (HIP events on the context's stream).
 * names: "msm_decompose","msm_sort","msm_accumulate","msm_reduce","k_acc_level1_g1","k_acc_level1_g2" (the
 * bucket-accumulation kernel alone, one launch per call),"msm_filter","ntt","pointwise","poseidon_leaf","poseidon_tree","r1cs_eval",
 * "solver_levels","rows_check","witgen_scatter","cex_commitments";
 * unknown names return 0.  calls = number of timed regions. */
double zkpor_phase_ms(zkpor_ctx* ctx, const char* name, uint64_t* calls);
/* counters of the context's last call, by name (0 for an unknown name): "msm_entries_w", "msm_entries_w_B", "msm_entries_w_K", "msm_entries_h" = the
 * sorted digit-stream entries (= bucket additions) of the last prove tail's A / B1 and B2 / K / Z multi-exponentiations */
int32_t zkpor_stat(zkpor_ctx* ctx, const char* name, uint64_t* value);
void zkpor_phase_reset(zkpor_ctx* ctx);

/* ---- proving key (replaces pk.UnsafeReadFrom's in-RAM key with an HBM-resident one) ------------------- */
int32_t zkpor_pk_create(zkpor_ctx* ctx, zkpor_pk** out);
void zkpor_pk_destroy(zkpor_pk* pk);
/* pts: n affine points exactly as gnark holds them (compacted: wires whose point is infinity are absent for
 * A and B; public/committed wires are absent for K).  Copied to HBM before returning. */
int32_t zkpor_pk_set_g1(zkpor_pk* pk, int which, const void* pts, size_t n);
int32_t zkpor_pk_set_g2(zkpor_pk* pk, int which, const void* pts, size_t n);
/* The same arrays in gnark-crypto's COMPRESSED encoding, as pk.WriteTo puts them on disk (src/keygen/main.go:46):
 * G1 = 32 bytes (big-endian X), G2 = 64 bytes (X.A1 | X.A0); the two top bits of the first byte are 01 infinity,
 * 10 / 11 compressed with the lexicographically smallest / largest Y.  Decompression (one square root per point) runs on
 * the device — the step pk.UnsafeReadFrom spends minutes of CPU time on (src/prover/prover/prover.go:336-349).
 * ZKPOR_E_ARG if an element is not a compressed point, has X >= p, or is not on the curve (last_error names it). */
int32_t zkpor_pk_set_g1_compressed(zkpor_pk* pk, int which, const uint8_t* compressed32, size_t n);
int32_t zkpor_pk_set_g2_compressed(zkpor_pk* pk, int which, const uint8_t* compressed64, size_t n);
/* stand-alone form: n compressed points -> n affine points in gnark's in-memory layout (host buffers) */
int32_t zkpor_g1_decompress(zkpor_ctx* ctx, const uint8_t* compressed32, size_t n, void* out_affine);
int32_t zkpor_g2_decompress(zkpor_ctx* ctx, const uint8_t* compressed64, size_t n, void* out_affine);
/* alpha,beta,delta: G1 affine (64 B each); beta2,delta2: G2 affine (128 B each).
 * inf_a/inf_b: n_wires bytes, non-zero where pk.InfinityA/B[i] is true (may be NULL = none).
 * committed_idx: the n_committed wire indices removed from K besides the public wires (may be NULL).
 * After this call the key arrays are re-laid out WIRE-INDEXED in HBM (infinity where gnark compacted), so one
 * sorted digit stream of the witness serves the A, B1, B2 and K multi-exponentiations. */
int32_t zkpor_pk_set_consts(zkpor_pk* pk, const void* alpha, const void* beta, const void* delta,
                            const void* beta2, const void* delta2, int log2_domain, const uint8_t* inf_a,
                            const uint8_t* inf_b, size_t n_wires, size_t n_public,
                            const uint32_t* committed_idx, size_t n_committed, int z_order);
/* ---- gnark key FILE: the container pk.WriteTo writes (src/keygen/main.go:46) and
 * pk.UnsafeReadFrom reads (src/prover/prover/prover.go:343).  Layout restated in csrc/keyfile.hip; all counts/offsets of
 * one stream, found by walking its headers on the host (no device needed for zkpor_pk_gnark_layout). */
typedef struct {
    uint64_t domain_cardinality;
    uint64_t n_a, n_b1, n_z, n_k, n_b2;                 /* points in gnark's compacted arrays */
    uint64_t off_alpha, off_a, off_b1, off_z, off_k;    /* byte offset of the first point of each section */
    uint64_t off_beta2, off_b2;
    uint64_t n_wires, n_inf_a, n_inf_b, off_inf_a, off_inf_b;
    uint64_t n_basis, off_basis, n_basis_sigma, off_basis_sigma; /* CommitmentKeys[0] */
    uint64_t bytes_total;
    uint32_t domain_header_bytes;                       /* 169, or 168 for streams without the withPrecompute byte */
    uint32_t n_commitment_keys;
} zkpor_pk_layout_t;
/* ZKPOR_E_ARG (+ text in err, if given) when the bytes are not a well-formed compressed bn254 Groth16 proving key:
 * every count is cross-checked (len(A)+NbInfinityA = nbWires, masks add up, the stream ends at its last byte ...). */
int32_t zkpor_pk_gnark_layout(const uint8_t* data, size_t len, zkpor_pk_layout_t* out, char* err, size_t err_len);
/* Load a whole key from the stream: arrays are decompressed on the device straight from `data` (e.g. the mapped file) and
 * re-laid out wire-indexed, exactly as after zkpor_pk_set_*_compressed + zkpor_pk_set_consts.  The stream does not say
 * which wires K leaves out, so the caller passes what the constraint system knows: n_public (ONE wire included) and the
 * committed + commitment wire indices (gnark r1cs.CommitmentInfo); len(K) must equal nbWires - n_public - n_committed.
 * z_order says how the file's G1.Z relates to the coefficient index of h (ZKPOR_Z_ORDER_*): the stream does not record it and the
 * two conventions load equally cleanly — a wrong choice yields proofs the verifier rejects, so it is the caller's statement about
 * the gnark version that wrote the key (bit-reversed at setup since gnark 0.9, which includes the fork pinned at go.mod:57-60;
 * natural before), to be confirmed once per key by verifying one proof.
 * info (may be NULL) receives the layout.  zkpor_pk_load_gnark maps `path` read-only and calls the _mem form. */
int32_t zkpor_pk_load_gnark_mem(zkpor_pk* pk, const uint8_t* data, size_t len, size_t n_public,
                                const uint32_t* committed_idx, size_t n_committed, int z_order, zkpor_pk_layout_t* info);
int32_t zkpor_pk_load_gnark(zkpor_pk* pk, const char* path, size_t n_public, const uint32_t* committed_idx,
                            size_t n_committed, int z_order, zkpor_pk_layout_t* info);
/* TEST/BENCH utility (no reference counterpart): fill every array of the key with valid curve points generated
 * on the device (random walks from seeded multiples of the generators).  Sizes follow SURVEY.md §8(d) C2 when
 * n_wires = 2^log2_domain.  The key is NOT a sound Groth16 key; it exercises the prover's data path at scale. */
int32_t zkpor_pk_synth(zkpor_pk* pk, int log2_domain, size_t n_wires, size_t n_public, size_t n_committed,
                       uint64_t seed);
/* TEST/BENCH utility: the same generator with a CIRCUIT's sparsity instead of the seeded one — A / B1 / B2 are infinity exactly where
 * inf_a / inf_b are non-zero (pk.InfinityA / InfinityB: a wire that appears in no L / R row of the constraint system), K at the public
 * wires and at removed_idx (the committed wires and the commitment wire: gnark r1cs.CommitmentInfo), the Pedersen bases hold n_basis points
 * (one per committed wire).  What the end-to-end runs of a compiled circuit prove against (bench.py `end_to_end`, tests/test_circuit_gpu.py). */
int32_t zkpor_pk_synth_masked(zkpor_pk* pk, int log2_domain, size_t n_wires, size_t n_public, const uint8_t* inf_a, const uint8_t* inf_b,
                              const uint32_t* removed_idx, size_t n_removed, size_t n_basis, uint64_t seed);
/* device pointer + length of a loaded (wire-indexed) array, for tests */
/* Shape of a loaded key: dims = { n_wires, n_public, n_committed, |Z|, log2_domain, msm_tables }.  What a host caller checks a
 * solver's output against before handing it over (the reference reads the same numbers off r1cs.GetNbConstraints() / the pk,
 * src/prover/prover/prover.go:317-349); works for keys held as fixed-base tables too. */
int32_t zkpor_pk_dims(zkpor_pk* pk, uint64_t dims[6]);
int32_t zkpor_pk_g1_dev(zkpor_pk* pk, int which, void** dev_ptr, size_t* n);
int32_t zkpor_pk_g2_dev(zkpor_pk* pk, int which, void** dev_ptr, size_t* n);

/* ---- multi-scalar multiplication (gnark-crypto G1Jac.MultiExp / G2Jac.MultiExp) ------------------------ */
/* generic: sum_i scalars[i]*points[i] over caller-provided arrays */
int32_t zkpor_msm_g1(zkpor_ctx* ctx, const void* points_affine, const uint64_t* scalars, size_t n, uint8_t out_jac[96]);
int32_t zkpor_msm_g2(zkpor_ctx* ctx, const void* points_affine, const uint64_t* scalars, size_t n, uint8_t out_jac[192]);
int32_t zkpor_msm_g1_dev(zkpor_ctx* ctx, const void* d_points, const void* d_scalars, size_t n, uint8_t out_jac[96]);
/* Test-facing: steps 1 + 2 of every multi-exponentiation alone (csrc/sort.hip; gnark-crypto's partitionScalars + bucket walk, call site
 * src/prover/prover/prover.go:269) — the signed-digit stream of n device-resident scalars, grouped by bucket, copied to HOST buffers of `cap`
 * entries each: keys_out[j] = (digit position % piece) * 2^(c-1) + |digit| - 1, ascending; vals_out[j] = ((scalar index * tables + digit position /
 * piece) << 1 | (digit negative)) | bit 30 if absent0[index] | bit 31 if absent1[index] (one byte per scalar, either may be NULL); zero digits are
 * dropped.  info = { entries, entries whose scalar is not in absent0, ... not in absent1, c, digits per scalar W, piece, 2^(c-1), sort levels }.
 * Window c: the context's "msm_window" or the automatic choice for (n, tables).  At most 2^27 scalars.  ZKPOR_E_ARG when cap is too small
 * (info[0] says what is needed). */
int32_t zkpor_msm_digits_dev(zkpor_ctx* ctx, const void* d_scalars, size_t n, int tables, const uint8_t* absent0, const uint8_t* absent1,
                             uint32_t* keys_out, uint32_t* vals_out, size_t cap, uint64_t info[8]);
int32_t zkpor_msm_g2_dev(zkpor_ctx* ctx, const void* d_points, const void* d_scalars, size_t n, uint8_t out_jac[192]);

/* sum of `count` Jacobian points on the HOST (no device needed): combines the partial results of a multi-GPU split of one
 * multi-exponentiation after an all-gather (RCCL has no user-defined reduction over curve points) */
int32_t zkpor_g1_jac_sum(const uint8_t* parts96, size_t count, uint8_t out_jac[96]);
int32_t zkpor_g2_jac_sum(const uint8_t* parts192, size_t count, uint8_t out_jac[192]);

/* ---- H polynomial (gnark computeH) ---------------------------------------------------------------------- */
/* a,b,c: n_constraints evaluations each (zero-padded to the domain internally); h_out: 2^log2_domain elements
 * in the order the key's Z expects (bit-reversed for ZKPOR_Z_ORDER_BITREV). */
int32_t zkpor_compute_h(zkpor_ctx* ctx, int log2_domain, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                        size_t n_constraints, uint64_t* h_out);
/* in-place on device: d_a, d_b, d_c hold 2^log2_domain elements (already zero-padded); h is left in d_a. */
int32_t zkpor_compute_h_dev(zkpor_ctx* ctx, int log2_domain, void* d_a, void* d_b, void* d_c);
/* single transforms (gnark-crypto fft.Domain.FFT / FFTInverse): decimation 0 = DIT, 1 = DIF */
int32_t zkpor_fft(zkpor_ctx* ctx, uint64_t* a, int log2n, int inverse, int decimation, int on_coset);
int32_t zkpor_fft_dev(zkpor_ctx* ctx, void* d_a, int log2n, int inverse, int decimation, int on_coset);

/* ---- Groth16 prove tail: everything in groth16.Prove after the R1CS solver ------------------------------ */
/* w: n_wires wire values (full assignment, ONE wire first); a,b,c: n_constraints evaluations; r,s: blinding
 * scalars (Montgomery Fr).  proof_out: Ar (G1 affine 64 B) | Bs (G2 affine 128 B) | Krs (G1 affine 64 B) as
 * Montgomery limbs.  Computes h, the A/B1/B2/K/Z multi-exponentiations and the r/s blinding on the device.
 *
 * BLINDING: r and s are the zero-knowledge randomness of the proof.  gnark draws them with fr.SetRandom (crypto/rand) inside
 * groth16.Prove; a caller of this ABI must do the same: uniform CSPRNG output, FRESH for every proof, canonical (< the
 * modulus; ZKPOR_E_ARG otherwise), never seeded or reused (reuse leaks linear relations between witnesses), and — for the
 * single-proof split — identical on all ranks and still secret.  The host-side multiplications by r and s are
 * variable-time double-and-add: do not run the prover where a co-tenant can time it.  zkpor_prove_tail_rand below draws
 * r and s itself from the operating system (getrandom) for callers without a CSPRNG at hand.
 *
 * HOST-POINTER FORM (what a cgo shim binds): w, a, b, c may be ordinary pageable memory (a Go slice).  The context keeps a
 * persistent staging area in HBM (no allocation per proof) and moves the vectors across PCIe itself ("copy_threads" above; direct
 * DMA if the range was page-locked with zkpor_host_register).  Nothing of the caller's memory is read after the call returns.
 * To hide the transfer keep two proofs in flight per GPU (two contexts, one caller each; host/prover_host.hpp does this).  The
 * callers of one GPU then TAKE TURNS on the device ("gpu_token"): a call that finds the GPU free sends w first and a, b, c
 * underneath its own A, B1, K sums (the shortest single proof); a call that finds another proof running moves all four vectors
 * across underneath that proof's kernels, waits for it to finish, and runs with everything resident.  Without the turns two
 * callers drift into lockstep — kernels sharing the GPU, finishing together, then both copying with the GPU idle.  Measured
 * (bench.py `boundary`): 97 % of the resident rate from pageable memory with two callers.  The turn is process-wide state (one
 * flag per GPU); zkpor_commit takes none. */
int32_t zkpor_prove_tail(zkpor_ctx* ctx, zkpor_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                         const uint64_t* c, size_t n_constraints, const uint64_t r[4], const uint64_t s[4],
                         uint8_t proof_out[256]);
/* the same, with r and s drawn inside the library from getrandom(2) and returned (Montgomery limbs) for the caller's records;
 * either output pointer may be NULL */
int32_t zkpor_prove_tail_rand(zkpor_ctx* ctx, zkpor_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                              const uint64_t* c, size_t n_constraints, uint64_t r_out[4], uint64_t s_out[4],
                              uint8_t proof_out[256]);
/* HOST-POINTER FORM WITH THE CONSTRAINT MATRICES RESIDENT (zkpor_r1cs_* below): only the wire vector crosses PCIe —
 * n_wires x 32 B per proof instead of (n_wires + 3 n_constraints) x 32 B (2.1 GB instead of 8.6 GB at 2^26) — and a, b, c =
 * L.w, R.w, O.w are evaluated in the staging area before computeH.  Replaces the same call as zkpor_prove_tail (groth16.Prove,
 * src/prover/prover/prover.go:269) for a caller that exported its constraint system once (go/export_r1cs).  `r1cs` may belong to
 * any context of the same GPU (the matrices are only read): two callers share one copy. */
typedef struct zkpor_r1cs zkpor_r1cs;
int32_t zkpor_prove_r1cs(zkpor_ctx* ctx, zkpor_pk* pk, zkpor_r1cs* r1cs, const uint64_t* w, const uint64_t r[4], const uint64_t s[4],
                         uint8_t proof_out[256]);
/* device-resident inputs; d_a/d_b/d_c must hold 2^log2_domain elements (zero padded) and are overwritten */
int32_t zkpor_prove_tail_dev(zkpor_ctx* ctx, zkpor_pk* pk, const void* d_w, void* d_a, void* d_b, void* d_c,
                             const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[256]);
/* the same with the inputs PRESERVED: d_a / d_b / d_c are only read (computeH's first pass reads them and writes the work buffers), d_wa / d_wb /
 * d_wc (2^log2_domain elements each, distinct from the inputs) are overwritten and d_wa holds h on return.  For a caller that proves
 * again from the same evaluations (another blinding, a retry) or keeps them for checking — gnark's computeH destroys its inputs as
 * zkpor_prove_tail_dev does, so this form has no counterpart there. */
int32_t zkpor_prove_tail_dev_keep(zkpor_ctx* ctx, zkpor_pk* pk, const void* d_w, const void* d_a, const void* d_b, const void* d_c, void* d_wa,
                                  void* d_wb, void* d_wc, const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[256]);
/* ---- single-proof split over several GPUs (BASELINE.json configs[4]: one 2^28 proof, 8 MI355X) ----------
 * Every GPU keeps one contiguous range of each key array and of the matching scalars, computes the five partial
 * multi-exponentiations, the partial sums (576 B per GPU) are all-gathered (RCCL; the only collective of the path besides
 * scattering h from the GPU that ran computeH), added with zkpor_g1/g2_jac_sum, and the proof is assembled on the host.
 * split.py shows the exchange; gnark has no counterpart (its MultiExp splits over CPU tasks). */
/* one rank's share straight from the key file: only wires [wire_lo, wire_hi) of A, B1, B2, K (located in gnark's compacted
 * arrays by counting the infinity / removed masks in front of the range) and points [z_lo, z_hi) of Z are uploaded and
 * decompressed, so no GPU ever holds more than its share; the result is a shard exactly as after zkpor_pk_keep_range.
 * z_order must be ZKPOR_Z_ORDER_BITREV (a range of the prover's order is contiguous only in such a file; for a natural-order key
 * load it whole and use zkpor_pk_keep_range). */
int32_t zkpor_pk_load_gnark_shard_mem(zkpor_pk* pk, const uint8_t* data, size_t len, size_t n_public,
                                      const uint32_t* committed_idx, size_t n_committed, size_t wire_lo, size_t wire_hi,
                                      size_t z_lo, size_t z_hi, int z_order, zkpor_pk_layout_t* info);
int32_t zkpor_pk_load_gnark_shard(zkpor_pk* pk, const char* path, size_t n_public, const uint32_t* committed_idx,
                                  size_t n_committed, size_t wire_lo, size_t wire_hi, size_t z_lo, size_t z_hi, int z_order,
                                  zkpor_pk_layout_t* info);
/* turn a loaded key into a shard: keep wires [wire_lo, wire_hi) of A, B1, B2, K (wire-indexed) and points [z_lo, z_hi) of Z
 * (in the order the prover's h has: bit-reversed), free the rest.  zkpor_prove_tail* then refuse the key (ZKPOR_E_STATE). */
int32_t zkpor_pk_keep_range(zkpor_pk* pk, size_t wire_lo, size_t wire_hi, size_t z_lo, size_t z_hi);
/* the five sums over the key's (or shard's) arrays: d_w = wire values of the kept wire range, d_h = h scalars of the kept Z
 * range (device, Montgomery Fr).  Either pointer may be NULL: then only the other half is computed and the missing sums are
 * infinity — a peer runs the four w-sums while the GPU that owns computeH is still busy and adds Z.h when its h block arrives.  sums_out = Jacobian points as
 * gnark-crypto holds them (X, Y, Z Montgomery limbs): A.w (96 B) | B1.w (96 B) | B2.w (192 B) | K.w (96 B) | Z.h (96 B). */
int32_t zkpor_prove_sums_dev(zkpor_ctx* ctx, zkpor_pk* pk, const void* d_w, const void* d_h, uint8_t sums_out[576]);
/* HOST ONLY (no device): blinding + assembly of the proof from the (already added) sums, as groth16.Prove does after its
 * MultiExps: Ar = alpha + A.w + r delta, Bs = beta2 + B2.w + s delta2, Krs = K.w + Z.h + s Ar + r Bs1 - rs delta. */
int32_t zkpor_prove_assemble(const void* alpha, const void* beta, const void* delta, const void* beta2, const void* delta2,
                             const uint8_t sums[576], const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[256]);
/* the five fixed points of a loaded key (G1 affine 64 B x3, G2 affine 128 B x2), e.g. to feed zkpor_prove_assemble */
int32_t zkpor_pk_consts(zkpor_pk* pk, void* alpha, void* beta, void* delta, void* beta2, void* delta2);

/* computeH itself spread over the same 2^world_log2 GPUs (csrc/ntt.hip ntt_shard_stage; index algebra in tools/ntt_model.py):
 * every rank holds 2^(log2_domain - world_log2) elements of a, b, c.  Each field pass of the transform is local under one of
 * two distributions of the memory position p — D_low (rank = low bits of p, local index p >> world_log2) or D_high (rank = top
 * bits, local index p mod local size) — so one all-to-all per transform replaces the NTT on a single GPU and the scatter of h:
 *   step 0 (a, b, c in D_low) | all-to-all to D_high | step 1 | all-to-all to D_low | step 2 (leaves the product in a) |
 *   all-to-all of a to D_high | step 3: a = this rank's contiguous block of h, in the order of the key's Z.
 * With "ntt_h" 1 (the default) the all-to-all after step 1 moves a and b ONLY: step 1 has taken c to its coefficients (times 1 / (N (g^N - 1))),
 * which stay in D_high; step 2 ignores d_c (may be NULL); step 3 needs d_c (ZKPOR_E_ARG without it) and subtracts it behind h's last pass —
 * six all-to-alls of a vector per proof.  With "ntt_h" 0 c goes to the coset like a and b (seven), and step 3 takes no c.
 * An all-to-all to D_high is: all_to_all_single of the local array (chunk d = elements [d M, (d+1) M), M = local size / W), then
 * zkpor_shard_transpose_dev(interleave = 1); to D_low: zkpor_shard_transpose_dev(interleave = 0), then all_to_all_single. */
int32_t zkpor_compute_h_shard_dev(zkpor_ctx* ctx, int log2_domain, int world_log2, int rank, void* d_a, void* d_b, void* d_c,
                                  int step);
/* out-of-place [W][M] -> [M][W] (interleave = 1: out[i W + r] = in[r M + i]) or back (interleave = 0) on 32-byte elements;
 * log2_local = log2 of the local array, W = 2^world_log2 */
int32_t zkpor_shard_transpose_dev(zkpor_ctx* ctx, void* d_out, const void* d_in, int log2_local, int world_log2, int interleave);

/* Pedersen commitment over the committed wires (gnark-crypto pedersen.ProvingKey.Commit / ProveKnowledge):
 * values: n_committed Fr; out: commitment | knowledge proof, G1 affine 64 B each */
int32_t zkpor_commit(zkpor_ctx* ctx, zkpor_pk* pk, const uint64_t* values, size_t n, uint8_t out_commit[64],
                     uint8_t out_pok[64]);
int32_t zkpor_commit_dev(zkpor_ctx* ctx, zkpor_pk* pk, const void* d_values, size_t n, uint8_t out_commit[64],
                         uint8_t out_pok[64]);
/* G1Affine.Marshal() (gnark-crypto: X | Y big-endian canonical, the identity as 0x40 then zeros) of one point this library returned
 * as Montgomery limbs — the bytes gnark hashes into the BSB22 challenge (constraint.SerializeCommitment; host/bsb22_challenge.hpp). */
int32_t zkpor_g1_marshal(const uint8_t affine_mont[64], uint8_t out_be[64]);
/* gnark raw proof bytes (proof.WriteRawTo): big-endian Ar.X|Ar.Y|Bs.X.A1|Bs.X.A0|Bs.Y.A1|Bs.Y.A0|Krs.X|Krs.Y|
 * u32 n_commitments | commitments... | pok  (388 B for one commitment; 324 B... for none) */
int32_t zkpor_proof_write_raw(const uint8_t proof[256], const uint8_t* commitments, uint32_t n_commitments,
                              const uint8_t pok[64], uint8_t* out, size_t out_cap, size_t* out_len);

/* ---- Poseidon account tree ------------------------------------------------------------------------------ */
/* one account, packed (mirrors utils.AccountInfo, src/utils/types.go:34-41) */
typedef struct {
    uint8_t id_be[32];        /* AccountId, 32 bytes big-endian (reduced mod r like PoseidonBytes does) */
    uint64_t equity[2];       /* TotalEquity, TotalDebt, TotalCollateral: < 2^128, little-endian words */
    uint64_t debt[2];
    uint64_t collateral[2];
    uint32_t n_assets;        /* number of entries in the asset array */
    uint32_t asset_off;       /* index of the first one */
} zkpor_account_t;            /* 88 bytes */
typedef struct {              /* utils.AccountAsset, src/utils/types.go:25-32; assets sorted by index */
    uint64_t equity, debt, loan, margin, portfolio_margin;
    uint32_t index, pad;
} zkpor_asset_t;              /* 48 bytes */
/* leaf hashes (utils.AccountInfoToHash) for n accounts padded to `tier` assets; out: n x 32 B big-endian */
int32_t zkpor_poseidon_leaves(zkpor_ctx* ctx, const zkpor_account_t* accounts, const zkpor_asset_t* assets,
                              size_t n_assets_total, size_t n, int tier, uint8_t* out32);
/* fixed-depth tree over n leaves (32 B big-endian each) at positions 0..n-1; every other leaf is nil_leaf.
 * levels_out (may be NULL): levels 1..depth concatenated, level l holding ceil(n/2^l) nodes, 32 B BE each. */
int32_t zkpor_merkle_build(zkpor_ctx* ctx, const uint8_t* leaves32_be, size_t n, int depth,
                           const uint8_t nil_leaf[32], uint8_t* levels_out, uint8_t root_out[32]);
/* device-resident leaves in Montgomery form (n x 32 B), root returned in Montgomery form: the bench path */
int32_t zkpor_merkle_build_dev(zkpor_ctx* ctx, const void* d_leaves_mont, size_t n, int depth,
                               const uint64_t nil_leaf_mont[4], uint64_t root_mont[4]);
/* ---- R1CS resident in HBM: a = L.w, b = R.w, c = O.w on the device (the evaluations computeH consumes) ----
 * gnark evaluates these inside groth16.Prove (src/prover/prover/prover.go:269) on the CPU; with the matrices resident only
 * the wire vector w has to cross PCIe per proof instead of a, b and c (3 x 2^26 x 32 B).  Shape follows gnark's compiled
 * system: a term is (coefficient id, wire id), coefficients sit in one shared table, each matrix is CSR over constraints.
 * The one-off export of (table, row_ptr, ids) from a gnark constraint system is sketched in INTEGRATION.md. */
typedef struct zkpor_r1cs zkpor_r1cs;
int32_t zkpor_r1cs_create(zkpor_ctx* ctx, size_t n_constraints, size_t n_wires, const uint64_t* coeff_table /* n_coeff x 4, Montgomery */,
                          size_t n_coeff, zkpor_r1cs** out);
void zkpor_r1cs_destroy(zkpor_r1cs* r1cs);
/* which: 0 = L (a), 1 = R (b), 2 = O (c); row_ptr has n_constraints + 1 entries, row_ptr[n_constraints] == nnz.
 * Indices are validated here (ZKPOR_E_ARG), so the kernel can trust them. */
int32_t zkpor_r1cs_set_matrix(zkpor_r1cs* r1cs, int which, const uint64_t* row_ptr, const uint32_t* coeff_ids,
                              const uint32_t* wire_ids, size_t nnz);
/* device buffers: w (n_wires Fr) in; a, b, c (domain_size Fr each) out, rows >= n_constraints written as zero (the
 * padding zkpor_compute_h_dev / zkpor_prove_tail_dev expect); asynchronous on the context's stream */
int32_t zkpor_r1cs_eval_dev(zkpor_r1cs* r1cs, const void* d_w, void* d_a, void* d_b, void* d_c, size_t domain_size);
/* the same on another context of the GPU the matrices live on (a second worker: its stream, its timers) */
int32_t zkpor_r1cs_eval_on(zkpor_ctx* ctx, zkpor_r1cs* r1cs, const void* d_w, void* d_a, void* d_b, void* d_c, size_t domain_size);
/* the solver's final check on the device: counts[0] = constraints with L.w * R.w != O.w for the wire vector d_w, counts[1] = the lowest
 * such row (2^64 - 1 when none).  What gnark's solver guarantees by construction has to be CHECKED when wires come from elsewhere (the
 * structured generators): a skipped instruction's constraint is only ever seen here.  Synchronous. */
int32_t zkpor_r1cs_check_dev(zkpor_r1cs* r1cs, const void* d_w, uint64_t counts[2]);
/* host buffers: a, b, c receive n_constraints elements each */
int32_t zkpor_r1cs_eval(zkpor_r1cs* r1cs, const uint64_t* w, uint64_t* a, uint64_t* b, uint64_t* c);

/* ---- CEX asset-list commitments and batch commitments (src/witness/witness/witness.go:159-198, utils.go:26-88,779-800) ----
 * One commitment = the chained Poseidon over 20 elements per asset (ConvertAssetInfoToBytes).  Prices and tier ratios are
 * constant over a run, the five running totals change per batch: pass the constants once and one totals row per CEX
 * state (before / after every batch); states are hashed independently, one per GPU thread. */
typedef struct { uint64_t boundary[2]; /* BoundaryValue, little-endian 128 bit, <= 2^118 */ uint8_t ratio; uint8_t pad[7]; } zkpor_tier_ratio_t;
typedef struct {               /* the constant part of utils.CexAssetInfo (src/utils/types.go:11-23), padded as PaddingTierRatios does */
    uint64_t base_price;
    zkpor_tier_ratio_t loan[12], margin[12], portfolio_margin[12];
} zkpor_cex_asset_const_t;     /* 872 bytes */
typedef struct { uint64_t total_equity, total_debt, loan_collateral, margin_collateral, portfolio_margin_collateral; } zkpor_cex_totals_t;
/* totals: n_states x n_assets rows (state-major); out32: n_states x 32 B big-endian.  n_assets = utils.AssetCounts (500)
 * in the reference, with the reserved slots already filled in (utils.go:781-792). */
int32_t zkpor_cex_commitments(zkpor_ctx* ctx, const zkpor_cex_asset_const_t* assets, size_t n_assets, const zkpor_cex_totals_t* totals,
                              size_t n_states, uint8_t* out32);
/* BatchCommitment = PoseidonBytes(AccountTreeRoot, Before, After, MinAccountIndex, MaxAccountIndex) for n batches */
int32_t zkpor_batch_commitments(zkpor_ctx* ctx, const uint8_t* roots32, const uint8_t* before32, const uint8_t* after32,
                                const uint32_t* min_index, const uint32_t* max_index, size_t n, uint8_t* out32);

/* Account totals and collateral tiers: fills equity / debt / collateral of every zkpor_account_t from its assets and the CEX asset
 * table — TotalEquity = sum equity_i * price_i, TotalDebt likewise, TotalCollateral = sum of the three tiered collateral values
 * (src/utils/utils.go:608-615, CalculateAssetValueForCollateral :648-661) — i.e. the big integers zkpor_poseidon_leaves then
 * hashes.  tier_info_out (may be NULL): 6 bytes per asset record = (loan index, loan flag, margin index, margin flag,
 * portfolio-margin index, flag), the claims calcAndSetCollateralInfo puts into the circuit witness (circuit/utils.go:227-278).
 * valid_out (may be NULL): 1 per account that passes the parser's checks (asset collateral <= equity, total collateral >=
 * total debt, no overflow), else 0. */
int32_t zkpor_account_totals(zkpor_ctx* ctx, zkpor_account_t* accounts, const zkpor_asset_t* assets, size_t n_assets_total, size_t n,
                             const zkpor_cex_asset_const_t* cex, size_t n_cex, uint8_t* tier_info_out, uint8_t* valid_out);

/* ---- FixedDepthMerkleTree (reference src/utils/merkletree/merkletree.go:27-52), resident in HBM ----
 * The two-phase usage of the reference: Set leaves (no hashing), Build (all internal nodes above a set leaf), then
 * Root / Get / GetProof.  Hashes cross the boundary as 32-byte big-endian canonical Fr, as the reference holds them.
 * Hasher = poseidon.NewPoseidon (the only one the reference passes: account_tree.go:14-23). */
typedef struct zkpor_tree zkpor_tree;
/* NewFixedDepthMerkleTree (:137-176): depth in [1,32], capacity <= 2^depth (ZKPOR_E_ARG where the reference panics) */
int32_t zkpor_tree_create(zkpor_ctx* ctx, int depth, const uint8_t nil_leaf[32], uint64_t capacity, zkpor_tree** out);
void zkpor_tree_destroy(zkpor_tree* tree);
/* nilHashes[level] (:159-170), level in [0, depth] */
int32_t zkpor_tree_nil_hash(zkpor_tree* tree, int level, uint8_t out[32]);
/* Set (:179-187) for n keys at once; a key >= capacity fails the whole call with ZKPOR_E_ARG and stores nothing */
int32_t zkpor_tree_set(zkpor_tree* tree, const uint32_t* keys, const uint8_t* values32_be, size_t n);
/* Set of the contiguous keys first_key .. first_key+n-1 from device-resident Montgomery leaves (zkpor_poseidon_leaves
 * output kept on the device, or the bench's synthetic leaves); asynchronous on the context's stream */
int32_t zkpor_tree_set_range_dev(zkpor_tree* tree, uint64_t first_key, const void* d_leaves_mont, size_t n);
/* Build (:192-279) */
int32_t zkpor_tree_build(zkpor_tree* tree);
/* Root (:282-284): reflects the last Build */
int32_t zkpor_tree_root(zkpor_tree* tree, uint8_t out[32]);
/* Get (:287-294) for n keys: the stored value, or nilHashes[0] for unset / out-of-capacity keys; out: n x 32 B */
int32_t zkpor_tree_get(zkpor_tree* tree, const uint32_t* keys, size_t n, uint8_t* out32);
/* GetProof (:297-308) for n keys: out = n x depth x 32 B, leaf-level sibling first; key >= 2^depth is ZKPOR_E_ARG */
int32_t zkpor_tree_get_proofs(zkpor_tree* tree, const uint32_t* keys, size_t n, uint8_t* out);
/* VerifyProof (:334-355) for n (key, leaf, proof) triples against one root; ok_out[i] = 1 / 0 */
int32_t zkpor_merkle_verify_proofs(zkpor_ctx* ctx, const uint8_t root[32], const uint32_t* keys, const uint8_t* proofs,
                                   const uint8_t* leaves32_be, size_t n, int depth, uint8_t* ok_out);
/* buildAccountTree (src/witness/main.go:130-199) for one chunk of accounts, device-resident end to end: (optionally) the account
 * totals from the CEX table, the leaf hashes, and Set at keys first_key .. first_key+n-1 — the 32-byte leaves never cross PCIe.
 * With cex != NULL the totals are computed first (zkpor_account_totals) and written back into `accounts`; with cex == NULL the
 * totals already in `accounts` are hashed.  Stream a large data set through this in chunks, then zkpor_tree_build once. */
int32_t zkpor_tree_set_accounts(zkpor_tree* tree, uint64_t first_key, zkpor_account_t* accounts, const zkpor_asset_t* assets,
                                size_t n_assets_total, size_t n, int tier, const zkpor_cex_asset_const_t* cex_or_null, size_t n_cex,
                                uint8_t* valid_out_or_null);
/* poseidon.Poseidon(inputs...) for `count` independent inputs of `len` elements each (Montgomery Fr in/out) */
int32_t zkpor_poseidon_hash(zkpor_ctx* ctx, const uint64_t* inputs, size_t len, size_t count, uint64_t* out);

/* ---- structured witness generation on the device ------------------------------------------------------------
 * r1cs.Solve (inside groth16.Prove, src/prover/prover/prover.go:269; hints registered at :68) fills the wire vector of
 * BatchCreateUserCircuit one instruction at a time on the host.  Two families of its wires are plain data-parallel functions of the
 * witness inputs and make up about half of the vector: the S-box wires of the in-circuit Poseidon gadget
 * (circuit/utils.go:12-21, 28-49; circuit/batch_create_user_circuit.go:104,129,181,270,281,320) and the 16-bit range-check limbs with
 * the multiplicities and inverse wires of the log-derivative lookup argument (circuit/batch_create_user_circuit.go:201-213,
 * circuit/utils.go:85-100,115-177).  These entry points compute them in HBM into "semantic slots"; the wire map that assigns a slot
 * to gnark's wire id comes from the one-off solver export (go/export_solver) and is applied by zkpor_witgen_scatter_dev.  The
 * remaining wires are solved by the levelized host executor (host/solver_exec.hpp).  All pointers are device pointers, all calls
 * asynchronous on the context's stream. */
/* S-boxes per permutation of width t: 8 t + R_P(t) (0 for widths the circuit does not use: anything but 3, 5, 6, 13) */
size_t zkpor_witgen_poseidon_sboxes(int t);
/* d_states: count x t Montgomery Fr, the initial states (state[0] = the capacity element), replaced by the final states.
 * d_trace: 3 * sboxes(t) * count Fr: trace[(s * 3 + c) * count + i] = for permutation i, S-box s in round order (full rounds: lanes
 * 0..t-1; partial rounds: lane 0), c = 0,1,2 -> x^2, x^4, x^5 — the three multiplication wires the gadget spends per S-box. */
int32_t zkpor_witgen_poseidon_trace_dev(zkpor_ctx* ctx, int t, void* d_states, size_t count, void* d_trace);
/* d_values: n Montgomery Fr below 2^(16 nb_limbs), nb_limbs in [1, 15].  d_limbs: nb_limbs * n Fr, limbs[l * n + i] = limb l of value i.
 * d_multiplicity: 65536 x u32 counters, incremented (not cleared) — the multiplicities of the 2^16-entry table.  d_bad: one u32,
 * incremented per value out of range. */
int32_t zkpor_witgen_limbs_dev(zkpor_ctx* ctx, const void* d_values, size_t n, int nb_limbs, void* d_limbs, void* d_multiplicity, void* d_bad);
/* d_out[i] = 1 / (challenge - d_values[i]) (the inverse wires of the log-derivative argument); a zero denominator gives 0 and
 * increments *d_bad */
int32_t zkpor_witgen_inverse_dev(zkpor_ctx* ctx, const void* d_values, size_t n, const uint64_t challenge[4], void* d_out, void* d_bad);
/* bit decompositions (api.ToBinary and the comparison gadgets; std/math/bits NBits): d_bits[b * n + i] = bit b of value i (Fr one / zero),
 * b < nbits <= 254; a value at or above 2^nbits increments *d_bad */
int32_t zkpor_witgen_bits_dev(zkpor_ctx* ctx, const void* d_values, size_t n, int nbits, void* d_bits, void* d_bad);
/* lookup results (logderivlookup.Table.Lookup: circuit/utils.go:137, circuit/batch_create_user_circuit.go:184-195,292): d_out[i] =
 * d_table[index_i], the index given as a field element (the query wire); an index outside the table yields 0 and increments *d_bad */
int32_t zkpor_witgen_gather_dev(zkpor_ctx* ctx, const void* d_table, size_t table_len, const void* d_indices, size_t n, void* d_out, void* d_bad);
/* circuit.IntegerDivision as checkAndGetIntegerDivisionRes calls it (circuit/utils.go:103-110,166-177; divisor = utils.PercentageMultiplier):
 * d_quotient[i], d_remainder[i] = DivMod(d_values[i], divisor) */
int32_t zkpor_witgen_divmod_small_dev(zkpor_ctx* ctx, const void* d_values, size_t n, uint32_t divisor, void* d_quotient, void* d_remainder);
/* d_w[d_wire_ids[i]] = d_src[i], i < n */
int32_t zkpor_witgen_scatter_dev(zkpor_ctx* ctx, void* d_w, const void* d_src, const uint32_t* d_wire_ids, size_t n);
/* the same, and d_known[d_wire_ids[i]] = 1: the generator's wires are handed to the solver program as already assigned (the d_known of
 * zkpor_solver_start_dev; n_wires bytes, cleared by the caller before the first scatter of a proof) */
int32_t zkpor_witgen_scatter_known_dev(zkpor_ctx* ctx, void* d_w, uint8_t* d_known, const void* d_src, const uint32_t* d_wire_ids, size_t n);

/* ---- the solver program on the device (the generic half) --------------------------------------------------------
 * r1cs.Solve inside groth16.Prove (src/prover/prover/prover.go:269; gnark constraint/bn254/solver.go, 3P) walks the compiled system's
 * levels — sets of mutually independent instructions: solve one constraint for its single unknown wire, or call a hint — with the host's
 * cores.  Here the exported program (go/export_solver -> the "ZKPSOLV" container, host/solver_file.hpp) sits in HBM next to the
 * constraint matrices (zkpor_r1cs_*) and a level is one launch with one GPU thread per instruction; runs of narrow levels are stepped
 * through by one workgroup.  Native hints: circuit.IntegerDivision (circuit/utils.go:103-110, registered at prover.go:68), NBits, InvZero,
 * DecomposeHint.  Any other hint (gnark's BSB22 commitment placeholder) is EXTERNAL: the run pauses in front of it and the caller serves it.
 * The wire vector never leaves the device: wires the structured generators (zkpor_witgen_*) produced are passed in as already known,
 * and the result feeds zkpor_r1cs_eval_dev / zkpor_commit_dev / zkpor_prove_tail_dev. */
typedef struct zkpor_solver zkpor_solver;
/* `r1cs` (all three matrices loaded) must outlive the solver; the container is copied and validated (ZKPOR_E_ARG).
 * THREADING: a solver's launches, phase timers and error text belong to ONE context — zkpor_solver_create binds it to the context its R1CS
 * was created on, zkpor_solver_create_on to any context of the same GPU.  A solver is single-caller like its context; a prover with two
 * workers per GPU creates one solver per worker context over ONE constraint-system handle (the matrices are only read; the program, ~2 GB at 2^26, is
 * copied per solver) and the workers solve side by side (tests/test_circuit_gpu.py). */
int32_t zkpor_solver_create(zkpor_r1cs* r1cs, const uint8_t* solver_container, size_t len, zkpor_solver** out);
int32_t zkpor_solver_create_on(zkpor_ctx* ctx, zkpor_r1cs* r1cs, const uint8_t* solver_container, size_t len, zkpor_solver** out);
void zkpor_solver_destroy(zkpor_solver* solver);
/* dims = {instructions, levels, constraint instructions, hint instructions, skipped instructions, levels holding an external hint,
 * kernel launches of the last run} */
int32_t zkpor_solver_dims(const zkpor_solver* solver, uint64_t dims[7]);
/* d_w: n_wires Montgomery Fr on the device, the first n_inputs (1 + nPublic + nSecret, gnark's order, wire 0 = ONE) filled.
 * d_known_or_null: n_wires bytes on the device, non-zero = this wire is already assigned (pre-filled by a generator); NULL = only the inputs.
 * Runs until the program ends (*paused_instr = 0xffffffff; every wire of d_w is then assigned, else ZKPOR_E_STATE) or until an external
 * hint is met (*paused_instr = its instruction index): read its inputs, provide its outputs, call zkpor_solver_resume_dev.  Synchronous.
 * Errors (ZKPOR_E_STATE, text in zkpor_last_error): the reference solver's — unsatisfied assertion, division by zero, a hint that
 * refuses its input (range check violated, zero divisor), an instruction with two unknown wires (wrong level order). */
int32_t zkpor_solver_start_dev(zkpor_solver* solver, void* d_w, size_t n_inputs, uint8_t* d_known_or_null, uint32_t* paused_instr);
int32_t zkpor_solver_resume_dev(zkpor_solver* solver, uint32_t* paused_instr);
/* a, b, c without evaluating the hash gadget's rows twice.  Two thirds of the terms of BatchCreateUserCircuit's matrices sit in the rows of its
 * Poseidon gadgets (14 to 79 terms each: the S-box input), and the solver's Poseidon instruction has exactly that value in a register.
 * zkpor_solver_set_abc_dev names the prove tail's a / b / c buffers (2^log2_domain elements each) for the runs that follow: the instruction
 * then writes a, b, c of its own rows (program container: firstRow); zkpor_solver_eval_abc_dev, called after the run, evaluates every other
 * row (zkpor_r1cs_eval_dev restricted to them) into the same buffers.  Bit-identical to zkpor_r1cs_eval_dev of the solved vector
 * (tests/test_circuit_gpu.py).  NULL pointers switch it off; ASYNC / prefetched instructions never write rows.
 * ASSERTIONS move with it.  A third of the real circuit's instructions only VERIFY a constraint whose wires are all assigned (container:
 * kind word bit 8, CHECK — host/solver_file.hpp).  gnark's solver evaluates each of them; a caller that computes a, b, c of every row right
 * after the run has the same information in a x b = c.  With the buffers set (and the context parameter `solver_defer_checks` at its default
 * 1) the run leaves the CHECK instructions out and zkpor_solver_eval_abc_dev verifies a x b = c on EVERY row after writing them (the rows
 * the Poseidon instructions wrote included): ZKPOR_E_STATE "N constraints are not satisfied, the first one is #row" — the reference
 * solver's error, reported one call later.  An unsatisfied assertion therefore no longer fails zkpor_solver_start_dev / resume_dev in
 * this mode; hints, lookups and divisions fail where they did.  The call is synchronous when it checks. */
int32_t zkpor_solver_set_abc_dev(zkpor_solver* solver, void* d_a, void* d_b, void* d_c);
int32_t zkpor_solver_eval_abc_dev(zkpor_solver* solver, const void* d_w, void* d_a, void* d_b, void* d_c, size_t domain_size);
/* Pipelining across proofs: starts the NEXT proof's long serial hash chains (instructions the program flags ASYNC — the two 10 000-element
 * CEX commitments of BatchCreateUserCircuit, circuit/batch_create_user_circuit.go:129,320: 834 chained permutations each, ~0.2 s of ONE wave)
 * on the solver's side stream while the current proof still runs its prove tail.  d_w_next: the next proof's wire vector with the assignment
 * already in place (wire 0 = ONE, public, secret); the next zkpor_solver_start_dev must be given exactly this pointer — it then skips those
 * instructions and joins the side stream in front of its last level (any other pointer abandons the prefetch).  One prefetch at a time, after
 * the current run has finished.  A prover loop has the next batch's witness row in hand while it proves the current one (prover.go:139-247). */
int32_t zkpor_solver_prefetch_dev(zkpor_solver* solver, void* d_w_next, size_t n_inputs);
/* the external hint the run is paused at: its evaluated input expressions (n_in x 4 limbs into in_values when not NULL) and its shape */
int32_t zkpor_solver_external_inputs(zkpor_solver* solver, uint32_t instr, uint64_t* in_values, size_t capacity, size_t* n_in, size_t* n_out);
/* the same into device memory (capacity >= n_in elements): the committed wires of a BSB22 commitment go straight to zkpor_commit_dev */
int32_t zkpor_solver_external_inputs_dev(zkpor_solver* solver, uint32_t instr, void* d_out, size_t capacity);
/* its output wires (n_out x 4 limbs, Montgomery), written into d_w and marked assigned */
int32_t zkpor_solver_external_outputs(zkpor_solver* solver, uint32_t instr, const uint64_t* out_values, size_t n_out);
/* groth16.Prove from the assigned inputs (prover.go:269 as one call): inputs cross PCIe, solver program -> w, r1cs -> a, b, c, prove tail,
 * all on the device.  For programs without external hints (else ZKPOR_E_STATE: drive the steps with the calls above + zkpor_commit_dev +
 * zkpor_r1cs_eval_dev + zkpor_prove_tail_dev).  `solver` must have been created on `r1cs`; any context of that GPU may call. */
int32_t zkpor_prove_inputs(zkpor_ctx* ctx, zkpor_pk* pk, zkpor_r1cs* r1cs, zkpor_solver* solver, const uint64_t* inputs, size_t n_inputs,
                           const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[256]);
/* host-buffer form for tests and small circuits: inputs in, w_out receives n_wires elements; (pre_ids, pre_vals) = wires assigned
 * elsewhere; stats = {constraint instructions, hint instructions, skipped instructions, kernel launches}.  External hints are refused. */
int32_t zkpor_solver_run(zkpor_solver* solver, const uint64_t* inputs, size_t n_inputs, const uint32_t* pre_ids, const uint64_t* pre_vals,
                         size_t n_pre, uint64_t* w_out, uint64_t stats[4]);

/* ---- device memory helpers for host languages without a HIP binding ------------------------------------ */
int32_t zkpor_dev_alloc(zkpor_ctx* ctx, size_t bytes, void** out);
int32_t zkpor_dev_free(zkpor_ctx* ctx, void* p);
int32_t zkpor_dev_upload(zkpor_ctx* ctx, void* dst, const void* src, size_t bytes);
int32_t zkpor_dev_download(zkpor_ctx* ctx, void* dst, const void* src, size_t bytes);
/* queued copy: returns immediately, the host range must stay valid until zkpor_sync (pin it with zkpor_host_register) */
int32_t zkpor_dev_upload_async(zkpor_ctx* ctx, void* dst, const void* src, size_t bytes);
/* page-lock / release a caller-owned host range (e.g. the backing array of a Go []fr.Element): uploads from pinned memory
 * run at PCIe rate and overlap with kernels */
int32_t zkpor_host_register(zkpor_ctx* ctx, void* ptr, size_t bytes);
int32_t zkpor_host_unregister(zkpor_ctx* ctx, void* ptr);
/* fill n Montgomery Fr elements with seeded pseudo-random values on the device. kind 0 = uniform,
 * kind 1 = the witness-like mixture of SURVEY.md §8(d) (25% {0,1}, 20% <2^16, 5% <2^64, 50% uniform) */
int32_t zkpor_dev_fill_fr(zkpor_ctx* ctx, void* d_out, size_t n, uint64_t seed, int kind);

/* out[i] = a[i]*b[i] (Montgomery Fr) and an async device-to-device copy, both on the context's stream */
int32_t zkpor_dev_fr_mul(zkpor_ctx* ctx, void* d_out, const void* d_a, const void* d_b, size_t n);
int32_t zkpor_dev_copy(zkpor_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
