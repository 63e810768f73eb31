package zkporgpu

// Prove is gnark's backend/groth16/bn254/prove.go with two substitutions (marked GPU below); the solver, the hints, the challenge
// hashing and the Proof struct stay gnark's, so the proof that comes out is read by the unmodified groth16.Verify
// (src/prover/prover/prover.go:276, src/verifier/main.go:284).  NOT COMPILED in the authoring image — go/README.md.

import (
	"errors"
	"hash"
	"math/big"

	"github.com/consensys/gnark-crypto/ecc"
	curve "github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark/backend"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	"github.com/consensys/gnark/backend/witness"
	"github.com/consensys/gnark/constraint"
	cs_bn254 "github.com/consensys/gnark/constraint/bn254"
	"github.com/consensys/gnark/constraint/solver"
	fcs "github.com/consensys/gnark/frontend/cs"
)

// bsb22ChallengeWith is the output of gnark's commitment hint (backend/groth16/bn254/prove.go): hash_to_field over the serialized
// commitment and the public / commitment wires committed next to it.  Shared by Prove and ProveOnDevice (solver.go).
func bsb22ChallengeWith(h hash.Hash, commitment *curve.G1Affine, hashed []*big.Int) fr.Element {
	h.Write(constraint.SerializeCommitment(commitment.Marshal(), hashed, (fr.Bits-1)/8+1))
	hashBts := h.Sum(nil)
	h.Reset()
	nbBuf := fr.Bytes
	if h.Size() < fr.Bytes {
		nbBuf = h.Size()
	}
	var res fr.Element
	res.SetBytes(hashBts[:nbBuf])
	return res
}

// Prove = groth16.Prove(r1cs, pk, fullWitness) on the GPU behind ctx.
func Prove(ctx *Context, r1cs *cs_bn254.R1CS, pk *ProvingKey, fullWitness witness.Witness, opts ...backend.ProverOption) (*groth16_bn254.Proof, error) {
	opt, err := backend.NewProverConfig(opts...)
	if err != nil {
		return nil, err
	}
	if opt.HashToFieldFn == nil {
		opt.HashToFieldFn = hashToField([]byte(constraint.CommitmentDst)) // gnark's default: hash_to_field with the BSB22 DST
	}
	commitmentInfo, _ := r1cs.CommitmentInfo.(constraint.Groth16Commitments)
	if len(commitmentInfo) > 1 {
		return nil, errors.New("zkporgpu: more than one commitment (BatchCreateUserCircuit has one)")
	}
	proof := &groth16_bn254.Proof{Commitments: make([]curve.G1Affine, len(commitmentInfo))}
	poks := make([]curve.G1Affine, len(commitmentInfo))

	solverOpts := opt.SolverOpts[:len(opt.SolverOpts):len(opt.SolverOpts)]
	bsb22ID := solver.GetHintID(fcs.Bsb22CommitmentComputePlaceholder)
	solverOpts = append(solverOpts, solver.OverrideHint(bsb22ID, func(_ *big.Int, in []*big.Int, out []*big.Int) error {
		i := int(in[0].Int64())
		in = in[1:]
		hashed := in[:len(commitmentInfo[i].PublicAndCommitmentCommitted)]
		committed := in[len(hashed):]
		values := make([]fr.Element, len(commitmentInfo[i].PrivateCommitted))
		for j, inJ := range committed {
			values[j].SetBigInt(inJ)
		}
		// GPU: pedersen Commit and ProveKnowledge share the digit stream of `values` — both sums in one call
		var e error
		if proof.Commitments[i], poks[i], e = ctx.Commit(pk, values); e != nil {
			return e
		}
		res := bsb22ChallengeWith(opt.HashToFieldFn, &proof.Commitments[i], hashed)
		res.BigInt(out[0])
		return nil
	}))

	_solution, err := r1cs.Solve(fullWitness, solverOpts...)
	if err != nil {
		return nil, err
	}
	solution := _solution.(*cs_bn254.R1CSSolution)
	wireValues := []fr.Element(solution.W)

	// fold the knowledge proofs with the challenge derived from the commitment wires (one commitment: the fold is poks[0])
	if len(commitmentInfo) > 0 {
		commitmentsSerialized := make([]byte, fr.Bytes*len(commitmentInfo))
		for i := range commitmentInfo {
			copy(commitmentsSerialized[fr.Bytes*i:], wireValues[commitmentInfo[i].CommitmentIndex].Marshal())
		}
		challenge, err := fr.Hash(commitmentsSerialized, []byte("G16-BSB22"), 1)
		if err != nil {
			return nil, err
		}
		if _, err = proof.CommitmentPok.Fold(poks, challenge[0], ecc.MultiExpConfig{NbTasks: 1}); err != nil {
			return nil, err
		}
	}

	// fresh blinding, as gnark samples it
	var r, s fr.Element
	if _, err = r.SetRandom(); err != nil {
		return nil, err
	}
	if _, err = s.SetRandom(); err != nil {
		return nil, err
	}
	// GPU: computeH + MultiExp x5 + deltas / Krs assembly
	proof.Ar, proof.Bs, proof.Krs, err = ctx.ProveTail(pk, wireValues, solution.A, solution.B, solution.C, &r, &s)
	if err != nil {
		return nil, err
	}
	return proof, nil
}
