// Copy to src/prover/prover/dispatcher_gpu.go (package prover).  NOT COMPILED in the authoring image — go/README.md.
//
// The in-process dispatcher BASELINE.json's north star names: ONE prover process drives every GPU of the node; the Redis list
// (BRPOP, prover.go:72-84) is replaced by a channel fed straight from the witness table.  Per GPU: ONE uploaded proving key
// (gpuKeys, loaded by the patched LoadSnarkParamsOnce) shared by `workersPerGPU` worker goroutines, each with its own
// zkporgpu.Context (HIP stream + workspace + staging).  Two workers per GPU is what the measured 97.5 % boundary rate needs
// (bench.py `boundary`: one proof's host->device copies hide under the other proof's kernels; the callers of one GPU take turns
// on the device inside the library) — two prover PROCESSES per GPU would need two copies of the key and do not fit.
//
// Semantics kept from Prover.Run (prover.go:139-247), as host/prover_host.hpp keeps them in C++ (tests/test_dispatcher_cpu.py,
// tests/test_dispatcher_gpu.py): exactly-once hand-out (status CAS Published -> Received, witness_model.go:129-152), the
// duplicate-proof guard (prover.go:208-225; a lost CreateProof race reads as "already proved"), Finished after the row is written,
// rerun = Received first, then Published, with an in-process claim so that N workers do not all take the latest row, and the
// process ends when no Published row is left.  Rows of another tier than the loaded one are parked until the feed is drained and then
// proved tier by tier (one key reload per tier instead of one per interleaving).  A tier change (LoadSnarkParamsOnce) stops the world: every worker finishes its
// proof, the keys of all GPUs are swapped, proving resumes.
package prover

import (
	"bytes"
	"encoding/base64"
	"encoding/json"
	"errors"
	"fmt"
	"sync"
	"time"

	"github.com/binance/zkmerkle-proof-of-solvency/circuit"
	"github.com/binance/zkmerkle-proof-of-solvency/src/prover/zkporgpu"
	"github.com/binance/zkmerkle-proof-of-solvency/src/utils"
	"github.com/binance/zkmerkle-proof-of-solvency/src/witness/witness"
	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/groth16"
	cs_bn254 "github.com/consensys/gnark/constraint/bn254"
	"github.com/consensys/gnark/frontend"
)

type gpuWorker struct {
	gpu int
	ctx *zkporgpu.Context
}

type dispatcher struct {
	p       *Prover
	tier    sync.RWMutex // readers: proofs in flight with the loaded tier; writer: LoadSnarkParamsOnce
	claimMu sync.Mutex
	claimed map[int64]bool // heights held by a worker of this process during a rerun
	failed  chan error
	// batches of ANOTHER tier than the loaded one are parked while the feed still hands out work (a tier switch reloads a multi-GB key
	// on every GPU; with tiers interleaved in the feed several workers would take turns reloading).  The reference never meets the case:
	// its single loop proves in height order and the witness service writes the tiers one after the other (witness.go:138-206).
	parkMu  sync.Mutex
	parked  map[int][]*witness.BatchWitness // by tier; rows already CASed to Received by this process
	drained bool                            // the feed has ended: parked tiers are proved now, one switch per tier
	workers []*gpuWorker                    // every worker of the process (set once, before the loops start): a tier switch trims their contexts
}

// RunInProcess is Prover.Run for one process that drives `gpus` with `workersPerGPU` workers each.
func (p *Prover) RunInProcess(rerun bool, gpus []int, workersPerGPU int) error {
	if len(gpus) == 0 || workersPerGPU < 1 {
		return errors.New("RunInProcess: need at least one GPU and one worker per GPU")
	}
	p.proofModel.CreateProofTable()
	p.GPUs = gpus
	d := &dispatcher{p: p, claimed: map[int64]bool{}, parked: map[int][]*witness.BatchWitness{}, failed: make(chan error, len(gpus)*workersPerGPU)}
	var workers []*gpuWorker
	for _, g := range gpus {
		for k := 0; k < workersPerGPU; k++ {
			ctx, err := zkporgpu.NewContext(g)
			if err != nil {
				return fmt.Errorf("GPU %d: %w", g, err)
			}
			defer ctx.Close()
			if workersPerGPU > 1 {
				// round 6: with several workers per GPU the prove tail runs on HIP streams with hardware queues of their own and takes the device turn — ONE
				// tail at a time, the other worker's solver program beside it, competing for compute units as they free up ("tail_streams"; no compute unit is
				// reserved any more: round 5's 32-unit reserve cost every tail 12.5 % of the machine, 339 -> 328 ms per zkpor50_1380 proof,
				// profiles/r06_ab_reserve_vs_own_queues.json).  It applies to the device-solver path only (ProveOnDevice: zkpor_solver_* ... zkpor_prove_tail_dev);
				// a host-pointer call (Prove: gnark's solver on the host, then zkpor_prove_tail) holds the turn over its whole device part and is never masked.
				if err := ctx.SetParam("tail_streams", 1); err != nil {
					return fmt.Errorf("GPU %d: %w", g, err)
				}
			}
			workers = append(workers, &gpuWorker{gpu: g, ctx: ctx})
		}
	}
	d.workers = workers
	heights := make(chan int64, 4*len(workers))
	if !rerun {
		go d.feed(heights) // replaces BRPOP: the queue IS the Published rows of the witness table
	} else {
		close(heights)
	}
	var wg sync.WaitGroup
	for _, w := range workers {
		wg.Add(1)
		go func(w *gpuWorker) {
			defer wg.Done()
			var err error
			if rerun {
				err = d.rerunLoop(w)
			} else {
				err = d.runLoop(w, heights)
			}
			if err != nil {
				d.failed <- err
			}
		}(w)
	}
	wg.Wait()
	select {
	case err := <-d.failed:
		return err
	default:
		fmt.Println("prover run finish...")
		return nil
	}
}

// feed hands out every Published height exactly once (what dbtool pushes to Redis in the reference).  It works on SNAPSHOTS of the
// Published set (paging with an offset over a set that shrinks while workers CAS rows to Received would skip rows): take all
// heights, hand out the ones not handed out before, repeat until a snapshot holds nothing new — rows the witness service publishes
// while the provers run are picked up by the next snapshot.
func (d *dispatcher) feed(out chan<- int64) {
	defer close(out)
	seen := map[int64]bool{}
	for {
		var snapshot []int64
		for offset := 0; ; offset += 4096 {
			hs, err := d.p.witnessModel.GetAllBatchHeightsByStatus(witness.StatusPublished, 4096, offset)
			if err == utils.DbErrQueryInterrupted || err == utils.DbErrQueryTimeout {
				time.Sleep(1 * time.Second)
				offset -= 4096
				continue
			}
			if err != nil || len(hs) == 0 {
				break
			}
			snapshot = append(snapshot, hs...)
		}
		fresh := 0
		for _, h := range snapshot {
			if !seen[h] {
				seen[h] = true
				fresh++
				out <- h
			}
		}
		if fresh == 0 { // "there is no published status witness in db, so quit"
			return
		}
	}
}

func (d *dispatcher) runLoop(w *gpuWorker, heights <-chan int64) error {
	for h := range heights {
		var rows []*witness.BatchWitness
		var err error
		for {
			rows, err = d.p.witnessModel.GetAndUpdateBatchesWitnessByHeight(int(h), witness.StatusPublished, witness.StatusReceived)
			if err == utils.DbErrQueryInterrupted || err == utils.DbErrQueryTimeout {
				time.Sleep(1 * time.Second)
				continue
			}
			break
		}
		if errors.Is(err, utils.DbErrNotFound) {
			continue // another prover (another node) won the CAS
		}
		if err != nil {
			return err
		}
		for _, bw := range rows {
			if d.park(bw) {
				continue
			}
			if err := d.proveAndStore(w, bw); err != nil {
				return err
			}
		}
	}
	// the feed is drained: the parked tiers, lowest first; every worker helps, the tier lock makes the switch happen once per tier
	d.parkMu.Lock()
	d.drained = true
	d.parkMu.Unlock()
	for {
		bw := d.unpark()
		if bw == nil {
			return nil
		}
		if err := d.proveAndStore(w, bw); err != nil {
			return err
		}
	}
}

// loadedTier reads Prover.CurrentSnarkParamsInUse under the tier lock: LoadSnarkParamsOnce writes it holding d.tier.Lock (ADVICE r04:
// park / unpark read it bare — a data race under `go test -race`)
func (d *dispatcher) loadedTier() int {
	d.tier.RLock()
	defer d.tier.RUnlock()
	return d.p.CurrentSnarkParamsInUse
}

// tierOfRow: the tier the prover would load for this row (decided by the first user: circuit.SetBatchCreateUserCircuitWitness :363-366)
func tierOfRow(bw *witness.BatchWitness) int {
	wc := utils.DecodeBatchWitness(bw.WitnessData)
	return utils.GetNonEmptyAssetsCountOfUser(wc.CreateUserOps[0].Assets)
}

// park keeps a row of another tier for later while the feed is still running; false = prove it now
func (d *dispatcher) park(bw *witness.BatchWitness) bool {
	cur := d.loadedTier()
	if cur == 0 { // nothing loaded yet: the first row decides
		return false
	}
	t := tierOfRow(bw)
	if t == cur {
		return false
	}
	d.parkMu.Lock()
	defer d.parkMu.Unlock()
	if d.drained {
		return false
	}
	d.parked[t] = append(d.parked[t], bw)
	return true
}

// unpark hands out the parked rows tier by tier (ascending), nil when none is left
func (d *dispatcher) unpark() *witness.BatchWitness {
	d.parkMu.Lock()
	defer d.parkMu.Unlock()
	best := -1
	for t, rows := range d.parked {
		if len(rows) > 0 && (best < 0 || t < best) {
			best = t
		}
	}
	if best < 0 {
		return nil
	}
	// rows of the tier that is loaded go first: the other workers are still proving them
	if cur := d.loadedTier(); len(d.parked[cur]) > 0 {
		best = cur
	}
	bw := d.parked[best][0]
	d.parked[best] = d.parked[best][1:]
	return bw
}

func (d *dispatcher) rerunLoop(w *gpuWorker) error {
	for {
		bw, err := d.claimLatest()
		if errors.Is(err, utils.DbErrNotFound) {
			fmt.Println("there is no received status witness in db, so quit")
			return nil
		}
		if err != nil {
			return err
		}
		err = d.proveAndStore(w, bw)
		d.claimMu.Lock()
		delete(d.claimed, bw.Height)
		d.claimMu.Unlock()
		if err != nil {
			return err
		}
	}
}

// claimLatest: FetchBatchWitnessForRerun (prover.go:107-137) made safe for several workers of one process.
func (d *dispatcher) claimLatest() (*witness.BatchWitness, error) {
	d.claimMu.Lock()
	defer d.claimMu.Unlock()
	for _, status := range []int64{witness.StatusReceived, witness.StatusPublished} {
		hs, err := d.p.witnessModel.GetAllBatchHeightsByStatus(status, 4096, 0)
		if err != nil && !errors.Is(err, utils.DbErrNotFound) {
			return nil, err
		}
		for i := len(hs) - 1; i >= 0; i-- { // latest first, as GetLatestBatchWitnessByStatus
			if d.claimed[hs[i]] {
				continue
			}
			bw, err := d.p.witnessModel.GetBatchWitnessByHeight(hs[i])
			if err != nil {
				return nil, err
			}
			d.claimed[hs[i]] = true
			return bw, nil
		}
	}
	return nil, utils.DbErrNotFound
}

// proveAndStore is the body of Run's inner loop (prover.go:178-245) for one row on one worker.
func (d *dispatcher) proveAndStore(w *gpuWorker, bw *witness.BatchWitness) error {
	p := d.p
	wc := utils.DecodeBatchWitness(bw.WitnessData)
	commitments, err := json.Marshal([][]byte{wc.BeforeCEXAssetsCommitment, wc.AfterCEXAssetsCommitment})
	if err != nil {
		return err
	}
	roots, err := json.Marshal([][]byte{wc.AccountTreeRoot})
	if err != nil {
		return err
	}
	proof, assetsCount, err := d.generateAndVerifyProof(w, wc, bw.Height)
	if err != nil {
		return fmt.Errorf("generate and verify proof error: %w", err)
	}
	var buf bytes.Buffer
	if _, err = proof.WriteRawTo(&buf); err != nil {
		return err
	}
	for { // duplicate-proof guard
		_, err = p.proofModel.GetProofByBatchNumber(bw.Height)
		if err == utils.DbErrQueryInterrupted || err == utils.DbErrQueryTimeout {
			time.Sleep(1 * time.Second)
			continue
		}
		break
	}
	if err != nil {
		row := &Proof{
			ProofInfo:               base64.StdEncoding.EncodeToString(buf.Bytes()),
			BatchNumber:             bw.Height,
			CexAssetListCommitments: string(commitments),
			AccountTreeRoots:        string(roots),
			BatchCommitment:         base64.StdEncoding.EncodeToString(wc.BatchCommitment),
			MinAccountIndex:         wc.MinAccountIndex,
			MaxAccountIndex:         wc.MaxAccountIndex,
			AssetsCount:             assetsCount,
		}
		if err = p.proofModel.CreateProof(row); err != nil {
			if _, again := p.proofModel.GetProofByBatchNumber(bw.Height); again != nil { // not a lost race: a real failure
				return fmt.Errorf("create blockProof of height %d failed: %w", bw.Height, err)
			}
		}
	} else {
		fmt.Printf("blockProof of height %d exists\n", bw.Height)
	}
	if err = p.witnessModel.UpdateBatchWitnessStatus(bw, witness.StatusFinished); err != nil {
		fmt.Println("update witness error:", err.Error())
	}
	return nil
}

// generateAndVerifyProof = GenerateAndVerifyProof (prover.go:250-283) on one worker's context, under the tier lock.
func (d *dispatcher) generateAndVerifyProof(w *gpuWorker, bwit *utils.BatchCreateUserWitness, batchNumber int64) (groth16.Proof, int, error) {
	p := d.p
	start := time.Now()
	circuitWitness, _ := circuit.SetBatchCreateUserCircuitWitness(bwit)
	tier := len(circuitWitness.CreateUserOps[0].Assets)
	d.tier.RLock()
	for tier != p.CurrentSnarkParamsInUse {
		d.tier.RUnlock()
		d.tier.Lock() // every proof in flight has finished
		if tier != p.CurrentSnarkParamsInUse {
			// round 6: the old tier's scratch (digit streams, accumulation regions, NTT tables: 40-60 GB per worker context after 2^26 proofs) goes back to
			// the device before the new tier's key arrives beside the old one (zkpor_trim; the next proof re-creates what it needs)
			for _, ow := range d.workers {
				_ = ow.ctx.Trim()
			}
		}
		p.LoadSnarkParamsOnce(tier) // no-op if another worker switched meanwhile; uploads the tier's key to every GPU in p.GPUs
		d.tier.Unlock()
		d.tier.RLock()
	}
	defer d.tier.RUnlock()
	full, err := frontend.NewWitness(circuitWitness, ecc.BN254.ScalarField())
	if err != nil {
		return nil, 0, err
	}
	public, err := frontend.NewWitness(circuit.NewVerifyBatchCreateUserCircuit(bwit.BatchCommitment), ecc.BN254.ScalarField(), frontend.PublicOnly())
	if err != nil {
		return nil, 0, err
	}
	proof, err := zkporgpu.Prove(w.ctx, p.R1cs.(*cs_bn254.R1CS), p.gpuKeys[w.gpu], full)
	if err != nil {
		return nil, 0, err
	}
	fmt.Printf("batch %d: proof generation cost %d ms on GPU %d\n", batchNumber, time.Since(start).Milliseconds(), w.gpu)
	if err = groth16.Verify(proof, p.VerifyingKey, public); err != nil { // gnark's verifier, unchanged (prover.go:276)
		return nil, 0, err
	}
	return proof, tier, nil
}
