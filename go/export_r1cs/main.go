// export_r1cs: one-off exporter of a compiled gnark constraint system (.r1cs as written by src/keygen/main.go:62, read at
// src/prover/prover/prover.go:317-324) to the flat little-endian container that zkmerkle-proof-of-solvency_amd/host/r1cs_file.hpp
// maps and feeds to zkpor_r1cs_* (SURVEY.md §8 f1): after that, a, b, c = L.w, R.w, O.w are evaluated in HBM and only the wire
// vector crosses PCIe per proof.  NOT COMPILED in the authoring image (no Go toolchain) — go/README.md.
//
//	go run ./export_r1cs zkpor50_1380.r1cs zkpor50_1380.zkr1cs
//
// Container (all integers little-endian):
//
//	magic "ZKPR1CS\x01"
//	u64 nConstraints, nWires, nPublic (ONE wire included), nSecret, nCoeff, nnzL, nnzR, nnzO, nCommitments
//	per commitment: u64 commitmentIndex, u64 nPrivate, u64 nPublicAndCommitment; u32 private[nPrivate]; u32 public[nPublicAndCommitment]
//	(pad to 8 bytes)
//	coefficient table: nCoeff x 4 x u64 (fr.Element limbs, Montgomery — exactly gnark's memory)
//	for L, R, O: u64 rowPtr[nConstraints+1]; u32 coeffID[nnz]; u32 wireID[nnz]; (pad to 8 bytes)
package main

import (
	"bufio"
	"encoding/binary"
	"fmt"
	"os"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/constraint"
	cs_bn254 "github.com/consensys/gnark/constraint/bn254"
)

func main() {
	if len(os.Args) != 3 {
		fmt.Println("usage: export_r1cs <in.r1cs> <out.zkr1cs>")
		os.Exit(2)
	}
	in, err := os.Open(os.Args[1])
	if err != nil {
		panic(err)
	}
	defer in.Close()
	ccs := groth16.NewCS(ecc.BN254)
	if _, err = ccs.ReadFrom(in); err != nil {
		panic(err)
	}
	r1cs := ccs.(*cs_bn254.R1CS)
	rows := r1cs.GetR1Cs()
	nWires := r1cs.GetNbPublicVariables() + r1cs.GetNbSecretVariables() + r1cs.GetNbInternalVariables()

	outF, err := os.Create(os.Args[2])
	if err != nil {
		panic(err)
	}
	defer outF.Close()
	w := bufio.NewWriterSize(outF, 1<<24)
	defer w.Flush()
	written := 0
	put := func(v interface{}) {
		if err := binary.Write(w, binary.LittleEndian, v); err != nil {
			panic(err)
		}
		written += binary.Size(v)
	}
	pad := func() {
		for written%8 != 0 {
			put(uint8(0))
		}
	}

	sides := []func(constraint.R1C) constraint.LinearExpression{
		func(r constraint.R1C) constraint.LinearExpression { return r.L },
		func(r constraint.R1C) constraint.LinearExpression { return r.R },
		func(r constraint.R1C) constraint.LinearExpression { return r.O },
	}
	var nnz [3]uint64
	for _, r := range rows {
		for s, sel := range sides {
			nnz[s] += uint64(len(sel(r)))
		}
	}
	info, _ := r1cs.CommitmentInfo.(constraint.Groth16Commitments)

	w.WriteString("ZKPR1CS\x01")
	written += 8
	put([]uint64{uint64(len(rows)), uint64(nWires), uint64(r1cs.GetNbPublicVariables()), uint64(r1cs.GetNbSecretVariables()),
		uint64(len(r1cs.Coefficients)), nnz[0], nnz[1], nnz[2], uint64(len(info))})
	for i := range info {
		put([]uint64{uint64(info[i].CommitmentIndex), uint64(len(info[i].PrivateCommitted)), uint64(len(info[i].PublicAndCommitmentCommitted))})
		for _, x := range info[i].PrivateCommitted {
			put(uint32(x))
		}
		for _, x := range info[i].PublicAndCommitmentCommitted {
			put(uint32(x))
		}
	}
	pad()
	for i := range r1cs.Coefficients {
		put([4]uint64(r1cs.Coefficients[i])) // fr.Element = [4]uint64, Montgomery
	}
	for s, sel := range sides {
		rowPtr := make([]uint64, len(rows)+1)
		cids := make([]uint32, 0, nnz[s])
		wids := make([]uint32, 0, nnz[s])
		for i, r := range rows {
			for _, t := range sel(r) {
				cids = append(cids, uint32(t.CoeffID()))
				wids = append(wids, uint32(t.WireID()))
			}
			rowPtr[i+1] = uint64(len(cids))
		}
		put(rowPtr)
		put(cids)
		put(wids)
		pad()
	}
	fmt.Printf("constraints %d, wires %d (public %d), coefficients %d, terms L/R/O %d/%d/%d, commitments %d, %d bytes\n",
		len(rows), nWires, r1cs.GetNbPublicVariables(), len(r1cs.Coefficients), nnz[0], nnz[1], nnz[2], len(info), written)
}
