// Package zkporgpu binds libzkpor.so (include/zkpor.h) — the MI355X backend of the Groth16 prove tail and the Poseidon account
// tree — into the reference.  NOT COMPILED in the authoring image (no Go toolchain); see go/README.md.
//
// Threading: a Context is single-caller (one call at a time) but may be used from any goroutine / OS thread — the library binds
// every call to the context's GPU itself.  For throughput run TWO contexts per GPU, one worker goroutine each: one proof's
// host->device copies then hide under the other proof's kernels (bench.py `boundary`, host/prover_host.hpp).
package zkporgpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -lzkpor
#include <stdlib.h>
#include "zkpor.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"unsafe"

	curve "github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	"github.com/consensys/gnark/constraint"
	cs_bn254 "github.com/consensys/gnark/constraint/bn254"
)

// Context is one HIP stream + workspace on one GPU.
type Context struct{ h *C.zkpor_ctx }

// NewContext fails when no usable gfx950 device exists: there is no CPU fallback.
func NewContext(device int) (*Context, error) {
	var h *C.zkpor_ctx
	if rc := C.zkpor_init(C.int(device), nil, &h); rc != 0 {
		return nil, fmt.Errorf("zkpor_init(%d) failed with %d: no usable gfx950 device (there is no CPU fallback)", device, int(rc))
	}
	return &Context{h}, nil
}

// SetParam sets a tuning knob of the context (zkpor_set_param).  A production prover on a 288 GB part sets "msm_tables" to 4 BEFORE
// loading the key: the key is then held as four fixed-base tables per point (112 GB at 2^26) and every proof needs 6 % fewer
// bucket additions.
func (c *Context) SetParam(name string, value int64) error {
	cname := C.CString(name)
	defer C.free(unsafe.Pointer(cname))
	return c.err(C.zkpor_set_param(c.h, cname, C.int64_t(value)))
}

// Trim hands the context's grow-only scratch (multi-exponentiation workspace, staging area, NTT tables: 40-60 GB after 2^26 proofs) back to
// the device (zkpor_trim); the next proof re-creates what it needs.  LoadSnarkParamsOnce calls it on every worker context before it uploads
// another tier's key, so that the old tier's scratch does not sit beside the new key.
func (c *Context) Trim() error { return c.err(C.zkpor_trim(c.h)) }

func (c *Context) Close() {
	if c.h != nil {
		C.zkpor_destroy(c.h)
		c.h = nil
	}
}

func (c *Context) err(rc C.int32_t) error {
	if rc == 0 {
		return nil
	}
	return errors.New(C.GoString(C.zkpor_last_error(c.h)))
}

// ProvingKey mirrors the shape of gnark's icicle backend key: the CPU key (still needed by the solver-side hint: its
// CommitmentKeys carry the Pedersen bases' lengths) plus its HBM-resident, wire-indexed copy.
type ProvingKey struct {
	*groth16_bn254.ProvingKey
	c   *Context
	dev *C.zkpor_pk
}

func (k *ProvingKey) Close() {
	if k.dev != nil {
		C.zkpor_pk_destroy(k.dev)
		k.dev = nil
	}
}

// CommittedWires lists the wires gnark leaves out of pk.G1.K besides the public ones: every privately committed wire and the
// commitment wires themselves (constraint.Groth16Commitments).
func CommittedWires(r1cs *cs_bn254.R1CS) []uint32 {
	info, ok := r1cs.CommitmentInfo.(constraint.Groth16Commitments)
	if !ok {
		return nil
	}
	var out []uint32
	for i := range info {
		for _, w := range info[i].PrivateCommitted {
			out = append(out, uint32(w))
		}
		out = append(out, uint32(info[i].CommitmentIndex))
	}
	return out
}

func boolsToBytes(b []bool) []byte {
	out := make([]byte, len(b)+1) // +1: never hand cgo the address of an empty slice
	for i, v := range b {
		if v {
			out[i] = 1
		}
	}
	return out
}

func u32ptr(s []uint32) *C.uint32_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&s[0]))
}

// Upload copies a key that gnark has already read (pk.UnsafeReadFrom, prover.go:343) into HBM, once per tier.  Slices are
// passed as gnark holds them in memory (Montgomery limbs); nothing is retained after return (cgo pointer rule).
func (c *Context) Upload(pk *groth16_bn254.ProvingKey, r1cs *cs_bn254.R1CS) (*ProvingKey, error) {
	out := &ProvingKey{ProvingKey: pk, c: c}
	if e := c.err(C.zkpor_pk_create(c.h, &out.dev)); e != nil {
		return nil, e
	}
	g1 := func(which C.int, pts []curve.G1Affine) error {
		var p unsafe.Pointer
		if len(pts) > 0 {
			p = unsafe.Pointer(&pts[0])
		}
		return c.err(C.zkpor_pk_set_g1(out.dev, which, p, C.size_t(len(pts))))
	}
	if e := g1(C.ZKPOR_G1_A, pk.G1.A); e != nil {
		return nil, e
	}
	if e := g1(C.ZKPOR_G1_B, pk.G1.B); e != nil {
		return nil, e
	}
	if e := g1(C.ZKPOR_G1_K, pk.G1.K); e != nil {
		return nil, e
	}
	sizeH := int(pk.Domain.Cardinality - 1) // the prover uses h[:Cardinality-1] against pk.G1.Z
	if e := g1(C.ZKPOR_G1_Z, pk.G1.Z[:sizeH]); e != nil {
		return nil, e
	}
	if len(pk.CommitmentKeys) > 1 {
		return nil, errors.New("zkporgpu: more than one commitment key (BatchCreateUserCircuit has one)")
	}
	if len(pk.CommitmentKeys) == 1 {
		ck := pk.CommitmentKeys[0]
		if e := g1(C.ZKPOR_G1_COMMIT_BASIS, ck.Basis); e != nil {
			return nil, e
		}
		if e := g1(C.ZKPOR_G1_COMMIT_BASIS_SIGMA, ck.BasisExpSigma); e != nil {
			return nil, e
		}
	}
	var b2 unsafe.Pointer
	if len(pk.G2.B) > 0 {
		b2 = unsafe.Pointer(&pk.G2.B[0])
	}
	if e := c.err(C.zkpor_pk_set_g2(out.dev, C.ZKPOR_G2_B, b2, C.size_t(len(pk.G2.B)))); e != nil {
		return nil, e
	}
	infA, infB := boolsToBytes(pk.InfinityA), boolsToBytes(pk.InfinityB)
	log2 := 0
	for (uint64(1) << log2) < pk.Domain.Cardinality {
		log2++
	}
	committed := CommittedWires(r1cs)
	return out, c.err(C.zkpor_pk_set_consts(out.dev,
		unsafe.Pointer(&pk.G1.Alpha), unsafe.Pointer(&pk.G1.Beta), unsafe.Pointer(&pk.G1.Delta),
		unsafe.Pointer(&pk.G2.Beta), unsafe.Pointer(&pk.G2.Delta), C.int(log2),
		(*C.uint8_t)(unsafe.Pointer(&infA[0])), (*C.uint8_t)(unsafe.Pointer(&infB[0])),
		C.size_t(len(pk.InfinityA)), C.size_t(r1cs.GetNbPublicVariables()),
		u32ptr(committed), C.size_t(len(committed)),
		C.ZKPOR_Z_ORDER_BITREV /* gnark >= 0.9 bit-reverses pk.G1.Z at setup; ZKPOR_Z_ORDER_NATURAL for older keys */))
}

// LoadFile skips gnark's reader: the library maps the .pk that keygen wrote (src/keygen/main.go:46), checks the framing,
// decompresses every array on the device (the minutes pk.UnsafeReadFrom spends on CPU square roots, prover.go:336-349) and lays
// the key out wire-indexed.  cpuKey may be nil when the caller needs no CPU copy.
func (c *Context) LoadFile(path string, r1cs *cs_bn254.R1CS, cpuKey *groth16_bn254.ProvingKey) (*ProvingKey, error) {
	out := &ProvingKey{ProvingKey: cpuKey, c: c}
	if e := c.err(C.zkpor_pk_create(c.h, &out.dev)); e != nil {
		return nil, e
	}
	committed := CommittedWires(r1cs)
	cpath := C.CString(path)
	defer C.free(unsafe.Pointer(cpath))
	return out, c.err(C.zkpor_pk_load_gnark(out.dev, cpath, C.size_t(r1cs.GetNbPublicVariables()), u32ptr(committed),
		C.size_t(len(committed)), C.ZKPOR_Z_ORDER_BITREV, nil))
}

// ProveTail is everything groth16.Prove does after the solver: h, the five MultiExps, blinding.  w = full wire assignment,
// a, b, c = the constraint evaluations (solution.A/B/C), r, s = fresh fr.SetRandom() values (NEVER reused, zkpor.h).
// The slices are ordinary Go heap memory: the library stages them across PCIe itself (persistent HBM staging, pinned bounce
// buffers) and reads nothing after returning.
func (c *Context) ProveTail(pk *ProvingKey, w, a, b, cc []fr.Element, r, s *fr.Element) (ar curve.G1Affine, bs curve.G2Affine, krs curve.G1Affine, err error) {
	var out [256]byte
	err = c.err(C.zkpor_prove_tail(c.h, pk.dev,
		(*C.uint64_t)(unsafe.Pointer(&w[0])), (*C.uint64_t)(unsafe.Pointer(&a[0])),
		(*C.uint64_t)(unsafe.Pointer(&b[0])), (*C.uint64_t)(unsafe.Pointer(&cc[0])), C.size_t(len(a)),
		(*C.uint64_t)(unsafe.Pointer(r)), (*C.uint64_t)(unsafe.Pointer(s)), (*C.uint8_t)(unsafe.Pointer(&out[0]))))
	if err != nil {
		return
	}
	// the 256 bytes ARE the in-memory form of the three affine points (Montgomery limbs)
	ar = *(*curve.G1Affine)(unsafe.Pointer(&out[0]))
	bs = *(*curve.G2Affine)(unsafe.Pointer(&out[64]))
	krs = *(*curve.G1Affine)(unsafe.Pointer(&out[192]))
	return
}

// ProveR1CS is ProveTail for a caller whose constraint matrices are resident (UploadR1CS): only w crosses PCIe (n_wires x 32 B per
// proof instead of n_wires + 3 n_constraints); a, b, c = L.w, R.w, O.w are evaluated in HBM.  `m` may have been uploaded through any
// Context of the same GPU.
func (c *Context) ProveR1CS(pk *ProvingKey, m *R1CS, w []fr.Element, r, s *fr.Element) (ar curve.G1Affine, bs curve.G2Affine, krs curve.G1Affine, err error) {
	var out [256]byte
	err = c.err(C.zkpor_prove_r1cs(c.h, pk.dev, m.h, (*C.uint64_t)(unsafe.Pointer(&w[0])),
		(*C.uint64_t)(unsafe.Pointer(r)), (*C.uint64_t)(unsafe.Pointer(s)), (*C.uint8_t)(unsafe.Pointer(&out[0]))))
	if err != nil {
		return
	}
	ar = *(*curve.G1Affine)(unsafe.Pointer(&out[0]))
	bs = *(*curve.G2Affine)(unsafe.Pointer(&out[64]))
	krs = *(*curve.G1Affine)(unsafe.Pointer(&out[192]))
	return
}

// Commit replaces pedersen.ProvingKey.Commit and ProveKnowledge (two MultiExps over the committed values) in one call.
func (c *Context) Commit(pk *ProvingKey, values []fr.Element) (commitment, pok curve.G1Affine, err error) {
	var p *C.uint64_t
	if len(values) > 0 {
		p = (*C.uint64_t)(unsafe.Pointer(&values[0]))
	}
	err = c.err(C.zkpor_commit(c.h, pk.dev, p, C.size_t(len(values)),
		(*C.uint8_t)(unsafe.Pointer(&commitment)), (*C.uint8_t)(unsafe.Pointer(&pok))))
	return
}

// R1CS keeps the three constraint matrices in HBM (SURVEY.md §8 f1): per proof only w crosses PCIe.
type R1CS struct {
	c *Context
	h *C.zkpor_r1cs
}

// UploadR1CS flattens the compiled system: coefficient table + one CSR matrix per side.
func (c *Context) UploadR1CS(r1cs *cs_bn254.R1CS) (*R1CS, error) {
	coeffs := r1cs.Coefficients
	nbWires := r1cs.GetNbPublicVariables() + r1cs.GetNbSecretVariables() + r1cs.GetNbInternalVariables()
	var h *C.zkpor_r1cs
	if e := c.err(C.zkpor_r1cs_create(c.h, C.size_t(r1cs.GetNbConstraints()), C.size_t(nbWires),
		(*C.uint64_t)(unsafe.Pointer(&coeffs[0])), C.size_t(len(coeffs)), &h)); e != nil {
		return nil, e
	}
	rows := r1cs.GetR1Cs()
	sides := []func(constraint.R1C) constraint.LinearExpression{
		func(r constraint.R1C) constraint.LinearExpression { return r.L },
		func(r constraint.R1C) constraint.LinearExpression { return r.R },
		func(r constraint.R1C) constraint.LinearExpression { return r.O },
	}
	for which, sel := range sides {
		rowPtr := make([]uint64, len(rows)+1)
		var cids, wids []uint32
		for i, r := range rows {
			for _, t := range sel(r) {
				cids = append(cids, uint32(t.CoeffID()))
				wids = append(wids, uint32(t.WireID()))
			}
			rowPtr[i+1] = uint64(len(cids))
		}
		if e := c.err(C.zkpor_r1cs_set_matrix(h, C.int(which), (*C.uint64_t)(unsafe.Pointer(&rowPtr[0])), u32ptr(cids), u32ptr(wids),
			C.size_t(len(cids)))); e != nil {
			C.zkpor_r1cs_destroy(h)
			return nil, e
		}
	}
	return &R1CS{c, h}, nil
}
