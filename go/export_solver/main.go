// export_solver: one-off exporter of the SOLVER PROGRAM of a compiled gnark constraint system — what r1cs.Solve walks inside
// groth16.Prove (src/prover/prover/prover.go:269) — to the flat container (host/solver_file.hpp) that csrc/solver.hip executes ON THE
// DEVICE (zkpor_solver_*) and host/solver_exec.hpp on all host threads (SURVEY.md §8 f4).  Companion of export_r1cs (the matrices).  NOT COMPILED in the authoring image (no Go
// toolchain; written against bnb-chain/gnark v0.10.1-0.20240910145009-4b5261061f04, go.mod:57) — go/README.md.
//
//	go run ./export_solver zkpor50_1380.r1cs zkpor50_1380.zksolv
//
// gnark (constraint/core.go, 3P) stores a compiled system as
//	Instructions []PackedInstruction{BlueprintID, ConstraintOffset, WireOffset, StartCallData}, CallData []uint32,
//	Blueprints []Blueprint, Levels [][]int (instruction ids that are mutually independent)
// and the solver dispatches per blueprint: BlueprintGenericR1C -> solve the instruction's ONE constraint for its single unknown wire;
// BlueprintGenericHint -> decode a HintMapping{HintID, Inputs []LinearExpression, OutputRange} from the call data and call the hint;
// every other blueprint of the R1CS builder (e.g. the lookup blueprint of the bnb fork) is either a solver (BlueprintSolvable: it fills
// wires itself) or a hint carrier (BlueprintHint).  This exporter flattens that into:
//
//	magic "ZKPSOLV\x02"
//	u64 nInstructions, nLevels, nHintNames, nCallData
//	hint names: u32 length + bytes each; pad to 8
//	u32 kind[nInstructions]   0 = solve constraint arg (| 0x100 = CHECK: the constraint assigns no wire, it is only verified — found by replaying
//	                              gnark's own instruction tree below; an executor whose caller verifies a x b = c on every row may leave these out);
//	                          1 = hint, call data at arg; 2 = skipped (wires produced by a device generator);
//	                          3 = table lookup (gnark's BlueprintLookupHint: std/lookup/logderivlookup): call data blockOff, nbEntries, nQ,
//	                              firstOutWire, the nQ index expressions — the table's entries are written ONCE per table (block at blockOff:
//	                              nEntries, entryOff[nEntries], the entry expressions), as gnark keeps them once per blueprint;
//	                          4 = a whole poseidon.Poseidon(...) gadget call (host/solver_file.hpp) — NOT produced here: the gadget leaves no
//	                              trace in a compiled system (its S-boxes are plain R1C instructions).  A gadget-aware export needs the
//	                              frontend's cooperation (a marker around std/hash/poseidon calls recording input expressions + the first S-box
//	                              wire; -skip then covers the constraint instructions and one kind-4 instruction per call takes their place).
//	                              This repo's own compiler (host/circuit/frontend.hpp) emits kind 4 directly; without the marker the S-boxes run
//	                              as ~3 x 169 constraint instructions per permutation on the generic executor (correct, deep).
//	u32 arg[nInstructions]
//	u64 levelPtr[nLevels+1]; u32 levelInstr[...]; pad to 8
//	u32 callData[]: per hint  nameId, nIn, nOut, outWire[nOut], then per input  nTerms, (coeffId, wireId)[nTerms]
//
// Hint names are the registered function names (solver.GetHintName), e.g. "…/circuit.IntegerDivision", "…/std/math/bits.nBits",
// "…/constraint/solver.InvZeroHint", "…/std/rangecheck.DecomposeHint", "…/std/internal/logderivarg.countHint", "…/frontend/cs.Bsb22CommitmentComputePlaceholder"; the executors
// bind their native implementations by the LAST path element (csrc/solver_instr.cuh hint_kind_of_name, host/solver_exec.hpp HintRegistry).  A hint
// without one — the BSB22 placeholder — pauses the device run for the caller to serve (zkpor_solver_external_*); the host executor takes it as a closure the caller
// registers under that name (HintRegistry.by_name).  The wire families a device generator produces are marked with -skip (a file of instruction ids, one per line, written
// by the wire-map tool from the same compiled system); without it everything runs on the host.
package main

import (
	"bufio"
	"encoding/binary"
	"fmt"
	"os"
	"strconv"
	"strings"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/constraint"
	cs_bn254 "github.com/consensys/gnark/constraint/bn254"
	"github.com/consensys/gnark/constraint/solver"
)

func main() {
	if len(os.Args) < 3 {
		fmt.Println("usage: export_solver <in.r1cs> <out.zksolv> [-skip ids.txt]")
		os.Exit(2)
	}
	in, err := os.Open(os.Args[1])
	if err != nil {
		panic(err)
	}
	defer in.Close()
	ccs := groth16.NewCS(ecc.BN254)
	if _, err = ccs.ReadFrom(bufio.NewReaderSize(in, 1<<26)); err != nil {
		panic(err)
	}
	r1cs := ccs.(*cs_bn254.R1CS)
	skip := map[int]bool{}
	if len(os.Args) == 5 && os.Args[3] == "-skip" {
		f, err := os.Open(os.Args[4])
		if err != nil {
			panic(err)
		}
		sc := bufio.NewScanner(f)
		for sc.Scan() {
			if id, err := strconv.Atoi(strings.TrimSpace(sc.Text())); err == nil {
				skip[id] = true
			}
		}
		f.Close()
	}

	var names []string
	nameID := map[string]uint32{}
	intern := func(n string) uint32 {
		if i := strings.LastIndexAny(n, "./"); i >= 0 {
			n = n[i+1:]
		}
		if id, ok := nameID[n]; ok {
			return id
		}
		nameID[n] = uint32(len(names))
		names = append(names, n)
		return nameID[n]
	}

	nIns := len(r1cs.Instructions)
	// which constraint instructions assign no wire: the same walk gnark's builder does when it levelises (Blueprint.UpdateInstructionTree
	// inserts the wires an instruction solves into the tree; an R1C instruction that inserts none is an assertion)
	tree := &wireTree{level: make([]int32, r1cs.GetNbInternalVariables()+r1cs.GetNbPublicVariables()+r1cs.GetNbSecretVariables())}
	for i := range tree.level {
		tree.level[i] = -1
	}
	tree.inputs = r1cs.GetNbPublicVariables() + r1cs.GetNbSecretVariables()
	kind := make([]uint32, nIns)
	arg := make([]uint32, nIns)
	var callData []uint32
	tableBlock := map[constraint.BlueprintID]uint32{} // lookup blueprint -> offset of its entry block in callData
	for i, pi := range r1cs.Instructions {
		ins := pi.Unpack(&r1cs.System)
		bp := r1cs.Blueprints[pi.BlueprintID]
		switch b := bp.(type) {
		case constraint.BlueprintR1C: // BlueprintGenericR1C: exactly one constraint
			_ = b
			kind[i], arg[i] = 0, uint32(ins.ConstraintOffset)
			before := tree.inserted
			bp.UpdateInstructionTree(ins, tree)
			if tree.inserted == before {
				kind[i] |= 0x100 // CHECK (host/solver_file.hpp INSTR_CHECK)
			}
		case constraint.BlueprintHint: // BlueprintGenericHint and the hint-carrying blueprints of the std gadgets
			var hm constraint.HintMapping
			b.DecompressHint(&hm, ins)
			kind[i], arg[i] = 1, uint32(len(callData))
			nOut := hm.OutputRange.End - hm.OutputRange.Start
			callData = append(callData, intern(solver.GetHintName(hm.HintID)), uint32(len(hm.Inputs)), nOut)
			for w := hm.OutputRange.Start; w < hm.OutputRange.End; w++ {
				callData = append(callData, w)
			}
			for _, le := range hm.Inputs {
				callData = append(callData, uint32(len(le)))
				for _, t := range le {
					callData = append(callData, uint32(t.CoeffID()), uint32(t.WireID()))
				}
			}
		case *constraint.BlueprintLookupHint: // a BlueprintSolvable, not a hint carrier: Solve reads the entries + the queries and sets the outputs
			// ins.Calldata = [len, nbEntries visible to this lookup, nbQueries, the query expressions as (n, (cID, vID) x n)...]
			block, ok := tableBlock[pi.BlueprintID]
			if !ok {
				// the entries once per table: nEntries, entryOff[nEntries] (from the block's first word), then the expressions as gnark compressed them
				block = uint32(len(callData))
				tableBlock[pi.BlueprintID] = block
				ec := b.EntriesCalldata
				var offs []uint32
				for p := 0; p < len(ec); p += 1 + 2*int(ec[p]) {
					offs = append(offs, uint32(p))
				}
				callData = append(callData, uint32(len(offs)))
				for _, o := range offs {
					callData = append(callData, uint32(1+len(offs))+o)
				}
				callData = append(callData, ec...)
			}
			kind[i], arg[i] = 3, uint32(len(callData))
			callData = append(callData, block, ins.Calldata[1], ins.Calldata[2], ins.WireOffset)
			callData = append(callData, ins.Calldata[3:]...)
		default:
			panic(fmt.Sprintf("instruction %d: blueprint %T is neither an R1C, a hint carrier nor the lookup blueprint — extend the exporter", i, bp))
		}
		if _, isR1C := bp.(constraint.BlueprintR1C); !isR1C {
			bp.UpdateInstructionTree(ins, tree) // hints, lookups: their output wires enter the tree
		}
		if skip[i] {
			kind[i] = 2
		}
	}

	out, err := os.Create(os.Args[2])
	if err != nil {
		panic(err)
	}
	defer out.Close()
	w := bufio.NewWriterSize(out, 1<<26)
	defer w.Flush()
	n := 0
	put := func(v interface{}) {
		if err := binary.Write(w, binary.LittleEndian, v); err != nil {
			panic(err)
		}
		n += binary.Size(v)
	}
	pad := func() {
		for n%8 != 0 {
			put(uint8(0))
		}
	}
	nLevelEntries := 0
	for _, l := range r1cs.Levels {
		nLevelEntries += len(l)
	}
	w.WriteString("ZKPSOLV\x02")
	n += 8
	put([]uint64{uint64(nIns), uint64(len(r1cs.Levels)), uint64(len(names)), uint64(len(callData))})
	for _, s := range names {
		put(uint32(len(s)))
		w.WriteString(s)
		n += len(s)
	}
	pad()
	put(kind)
	put(arg)
	pad()
	ptr := make([]uint64, len(r1cs.Levels)+1)
	flat := make([]uint32, 0, nLevelEntries)
	for i, l := range r1cs.Levels {
		for _, id := range l {
			flat = append(flat, uint32(id))
		}
		ptr[i+1] = uint64(len(flat))
	}
	put(ptr)
	put(flat)
	pad()
	put(callData)
	fmt.Printf("%d instructions (%d skipped), %d levels, %d hint names %v, %d words of call data\n", nIns, len(skip), len(r1cs.Levels), len(names), names, len(callData))
}

// wireTree is the constraint.InstructionTree the blueprints update while gnark levelises a system.  gnark's contract (constraint/core.go,
// 3P-recalled — ADVICE r04): HasWire is FALSE for inputs and constants and TRUE for every internal wire; an internal wire no instruction
// has solved yet reports GetWireLevel == LevelUnset.  BlueprintGenericR1C.UpdateInstructionTree skips the wires without HasWire, takes
// the one internal wire that is still LevelUnset as the constraint's output and inserts it one level above its deepest solved operand —
// so an R1C instruction that inserts NO wire has no unknown left: an assertion (CHECK).  `inserted` counts InsertWire calls on internal
// wires only (round 4 had HasWire inverted: products of intermediates were flagged CHECK and assertions over inputs were not).
type wireTree struct {
	level    []int32 // per wire; -1 = LevelUnset
	inserted int
	inputs   int // 1 + nPublic + nSecret wires in front: never solved by an instruction
}

func (t *wireTree) InsertWire(wire uint32, level constraint.Level) {
	if int(wire) < t.inputs || int(wire) >= len(t.level) {
		return
	}
	t.level[wire] = int32(level)
	t.inserted++
}
func (t *wireTree) HasWire(wire uint32) bool { return int(wire) >= t.inputs && int(wire) < len(t.level) }
func (t *wireTree) GetWireLevel(wire uint32) constraint.Level {
	if !t.HasWire(wire) {
		return constraint.LevelUnset
	}
	return constraint.Level(t.level[wire]) // -1 == constraint.LevelUnset until inserted
}
