package zkporgpu

// The solver program on the GPU (include/zkpor.h zkpor_solver_*, csrc/solver.hip): r1cs.Solve of groth16.Prove
// (src/prover/prover/prover.go:269) runs in HBM next to the constraint matrices; only the ASSIGNED INPUTS of a batch cross PCIe.
// NOT COMPILED in the authoring image (no Go toolchain) — go/README.md.
//
// One-off per circuit:   go run ./export_solver  ->  <name>.solver   (levels, instructions, hint call data of the compiled system)
//                        m, _ := ctx.UploadR1CS(r1cs);  sp, _ := ctx.UploadSolver(m, solverBytes)
// Per batch:             proof, err := zkporgpu.ProveOnDevice(ctx, pk, m, sp, r1cs, fullWitness)

/*
#include <stdlib.h>
#include "zkpor.h"
*/
import "C"

import (
	"errors"
	"math/big"
	"unsafe"

	curve "github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	"github.com/consensys/gnark/backend/witness"
	"github.com/consensys/gnark/constraint"
	cs_bn254 "github.com/consensys/gnark/constraint/bn254"
)

const notPaused = 0xffffffff

// Solver is a compiled circuit's solver program resident on the GPU of the R1CS it was created on.  One run at a time.
type Solver struct {
	c *Context
	m *R1CS
	h *C.zkpor_solver
}

// UploadSolver copies the container written by go/export_solver to the device and validates it against the matrices.
func (c *Context) UploadSolver(m *R1CS, container []byte) (*Solver, error) {
	if len(container) == 0 {
		return nil, errors.New("zkporgpu: empty solver container")
	}
	var h *C.zkpor_solver
	if e := c.err(C.zkpor_solver_create(m.h, (*C.uint8_t)(unsafe.Pointer(&container[0])), C.size_t(len(container)), &h)); e != nil {
		return nil, e
	}
	return &Solver{c, m, h}, nil
}

func (s *Solver) Close() {
	if s.h != nil {
		C.zkpor_solver_destroy(s.h)
		s.h = nil
	}
}

// ProveInputs is groth16.Prove for a circuit WITHOUT a commitment: the assigned inputs (1, public, secret — the vector
// witness.Vector() holds behind the constant wire) go in, Ar / Bs / Krs come out; solver, a / b / c and the prove tail run on the device.
func (c *Context) ProveInputs(pk *ProvingKey, m *R1CS, sp *Solver, inputs []fr.Element, r, s *fr.Element) (ar curve.G1Affine, bs curve.G2Affine, krs curve.G1Affine, err error) {
	var out [256]byte
	err = c.err(C.zkpor_prove_inputs(c.h, pk.dev, m.h, sp.h, (*C.uint64_t)(unsafe.Pointer(&inputs[0])), C.size_t(len(inputs)),
		(*C.uint64_t)(unsafe.Pointer(r)), (*C.uint64_t)(unsafe.Pointer(s)), (*C.uint8_t)(unsafe.Pointer(&out[0]))))
	if err != nil {
		return
	}
	ar = *(*curve.G1Affine)(unsafe.Pointer(&out[0]))
	bs = *(*curve.G2Affine)(unsafe.Pointer(&out[64]))
	krs = *(*curve.G1Affine)(unsafe.Pointer(&out[192]))
	return
}

// ProveOnDevice is groth16.Prove for BatchCreateUserCircuit (one BSB22 commitment): the solver program runs on the device and PAUSES at
// gnark's commitment placeholder hint; the committed wires are read where they are (zkpor_solver_external_inputs_dev), committed
// (zkpor_commit_dev: Pedersen commitment + knowledge proof in one pass), the challenge is hashed on the host exactly as gnark's
// prove.go does (constraint.SerializeCommitment + hash_to_field with the BSB22 DST) and handed back as the hint's output; the run
// resumes, a / b / c are evaluated in HBM and the prove tail follows.  The wire vector never exists in host memory.
func ProveOnDevice(ctx *Context, pk *ProvingKey, m *R1CS, sp *Solver, r1cs *cs_bn254.R1CS, fullWitness witness.Witness) (*groth16_bn254.Proof, error) {
	commitmentInfo, _ := r1cs.CommitmentInfo.(constraint.Groth16Commitments)
	if len(commitmentInfo) > 1 {
		return nil, errors.New("zkporgpu: more than one commitment (BatchCreateUserCircuit has one)")
	}
	vec, ok := fullWitness.Vector().(fr.Vector)
	if !ok {
		return nil, errors.New("zkporgpu: witness is not over bn254's scalar field")
	}
	nWires := r1cs.NbInternalVariables + r1cs.GetNbPublicVariables() + r1cs.GetNbSecretVariables()
	domain := int(pk.Domain.Cardinality)
	// device buffers of this proof: w, and a / b / c at the domain's size (a pool in a long-running prover)
	dW, err := ctx.Alloc(nWires * fr.Bytes)
	if err != nil {
		return nil, err
	}
	defer ctx.Free(dW)
	var dABC [3]unsafe.Pointer
	for i := range dABC {
		if dABC[i], err = ctx.Alloc(domain * fr.Bytes); err != nil {
			return nil, err
		}
		defer ctx.Free(dABC[i])
	}
	inputs := make([]fr.Element, 1+len(vec)) // wire 0 = ONE, then public, then secret: gnark's wire order
	inputs[0].SetOne()
	copy(inputs[1:], vec)
	if err = ctx.UploadTo(dW, unsafe.Pointer(&inputs[0]), len(inputs)*fr.Bytes); err != nil {
		return nil, err
	}
	proof := &groth16_bn254.Proof{Commitments: make([]curve.G1Affine, len(commitmentInfo))}
	var pok curve.G1Affine
	var paused C.uint32_t
	// a, b, c named before the run: the Poseidon instructions write their own rows, the assertions (container flag CHECK) are left to the
	// a x b = c pass of zkpor_solver_eval_abc_dev below (include/zkpor.h; an unsatisfied constraint is reported there)
	if err = ctx.err(C.zkpor_solver_set_abc_dev(sp.h, dABC[0], dABC[1], dABC[2])); err != nil {
		return nil, err
	}
	if err = ctx.err(C.zkpor_solver_start_dev(sp.h, dW, C.size_t(len(inputs)), nil, &paused)); err != nil {
		return nil, err
	}
	for paused != notPaused {
		// the only external hint of this circuit: bsb22CommitmentComputePlaceholder(i, hashed..., committed...)
		var nIn, nOut C.size_t
		if err = ctx.err(C.zkpor_solver_external_inputs(sp.h, paused, nil, 0, &nIn, &nOut)); err != nil {
			return nil, err
		}
		if len(commitmentInfo) != 1 || nOut != 1 {
			return nil, errors.New("zkporgpu: the solver program stops at a hint this prover does not serve")
		}
		nHashed := len(commitmentInfo[0].PublicAndCommitmentCommitted)
		nCommitted := int(nIn) - 1 - nHashed
		dIn, e := ctx.Alloc(int(nIn) * fr.Bytes)
		if e != nil {
			return nil, e
		}
		if err = ctx.err(C.zkpor_solver_external_inputs_dev(sp.h, paused, dIn, nIn)); err == nil {
			committed := unsafe.Add(dIn, (1+nHashed)*fr.Bytes)
			err = ctx.err(C.zkpor_commit_dev(ctx.h, pk.dev, committed, C.size_t(nCommitted),
				(*C.uint8_t)(unsafe.Pointer(&proof.Commitments[0])), (*C.uint8_t)(unsafe.Pointer(&pok))))
		}
		hashed := make([]fr.Element, nHashed) // the public / commitment wires hashed next to the commitment (none in this circuit)
		if err == nil && nHashed > 0 {
			err = ctx.DownloadFrom(unsafe.Pointer(&hashed[0]), unsafe.Add(dIn, fr.Bytes), nHashed*fr.Bytes)
		}
		ctx.Free(dIn)
		if err != nil {
			return nil, err
		}
		hashedBig := make([]*big.Int, nHashed)
		for j := range hashed {
			hashedBig[j] = hashed[j].BigInt(new(big.Int))
		}
		// gnark's hashing of the hint, shared with Prove (prove.go of this package); the default hash_to_field with the BSB22 DST
		challenge := bsb22ChallengeWith(hashToField([]byte(constraint.CommitmentDst)), &proof.Commitments[0], hashedBig)
		if err = ctx.err(C.zkpor_solver_external_outputs(sp.h, paused, (*C.uint64_t)(unsafe.Pointer(&challenge)), 1)); err != nil {
			return nil, err
		}
		if err = ctx.err(C.zkpor_solver_resume_dev(sp.h, &paused)); err != nil {
			return nil, err
		}
	}
	if err = ctx.err(C.zkpor_solver_eval_abc_dev(sp.h, dW, dABC[0], dABC[1], dABC[2], C.size_t(domain))); err != nil {
		return nil, err // incl. "N constraints are not satisfied, the first one is #row"
	}
	var r, s fr.Element
	if _, err = r.SetRandom(); err != nil {
		return nil, err
	}
	if _, err = s.SetRandom(); err != nil {
		return nil, err
	}
	var out [256]byte
	if err = ctx.err(C.zkpor_prove_tail_dev(ctx.h, pk.dev, dW, dABC[0], dABC[1], dABC[2],
		(*C.uint64_t)(unsafe.Pointer(&r)), (*C.uint64_t)(unsafe.Pointer(&s)), (*C.uint8_t)(unsafe.Pointer(&out[0])))); err != nil {
		return nil, err
	}
	proof.Ar = *(*curve.G1Affine)(unsafe.Pointer(&out[0]))
	proof.Bs = *(*curve.G2Affine)(unsafe.Pointer(&out[64]))
	proof.Krs = *(*curve.G1Affine)(unsafe.Pointer(&out[192]))
	proof.CommitmentPok = pok
	return proof, nil
}

// device memory helpers over zkpor_dev_* (a Go program has no HIP binding of its own)
func (c *Context) Alloc(bytes int) (unsafe.Pointer, error) {
	var p unsafe.Pointer
	err := c.err(C.zkpor_dev_alloc(c.h, C.size_t(bytes), &p))
	return p, err
}
func (c *Context) Free(p unsafe.Pointer) { C.zkpor_dev_free(c.h, p) }
func (c *Context) UploadTo(dst, src unsafe.Pointer, bytes int) error {
	return c.err(C.zkpor_dev_upload(c.h, dst, src, C.size_t(bytes)))
}
func (c *Context) DownloadFrom(dst, src unsafe.Pointer, bytes int) error {
	return c.err(C.zkpor_dev_download(c.h, dst, src, C.size_t(bytes)))
}
