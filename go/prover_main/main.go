// Replacement for src/prover/main.go: the same flags and config, plus -gpus / -workers_per_gpu for the in-process dispatcher
// (src/prover/prover/dispatcher_gpu.go = go/prover_gpu/dispatcher_gpu.go).  Without -gpus it behaves as the reference's main:
// one serial Run loop (prover.go:139-247) on the GPU named by ZKPOR_GPU, fed by the Redis list.  NOT COMPILED in the authoring
// image — go/README.md.
package main

import (
	"encoding/json"
	"flag"
	"fmt"
	"io/ioutil"
	"strconv"
	"strings"

	"github.com/binance/zkmerkle-proof-of-solvency/src/prover/config"
	"github.com/binance/zkmerkle-proof-of-solvency/src/prover/prover"
	"github.com/binance/zkmerkle-proof-of-solvency/src/utils"
)

func main() {
	proverConfig := &config.Config{}
	content, err := ioutil.ReadFile("config/config.json")
	if err != nil {
		panic(err.Error())
	}
	if err = json.Unmarshal(content, proverConfig); err != nil {
		panic(err.Error())
	}
	if len(proverConfig.AssetsCountTiers) != len(proverConfig.ZkKeyName) {
		panic("asset tiers and asset tier names should have the same length")
	}
	remotePasswdConfig := flag.String("remote_password_config", "", "fetch password from aws secretsmanager")
	rerun := flag.Bool("rerun", false, "flag which indicates rerun proof generation")
	gpus := flag.String("gpus", "", "comma-separated GPU indices driven by THIS process through the in-process dispatcher, e.g. 0,1,2,3,4,5,6,7 (empty: the reference's serial loop on $ZKPOR_GPU)")
	workers := flag.Int("workers_per_gpu", 2, "worker goroutines (one zkporgpu.Context each) per GPU; 2 hides one proof's PCIe copies under the other's kernels")
	flag.Parse()
	if *remotePasswdConfig != "" {
		s, err := utils.GetMysqlSource(proverConfig.MysqlDataSource, *remotePasswdConfig)
		if err != nil {
			panic(err.Error())
		}
		proverConfig.MysqlDataSource = s
	}
	p := prover.NewProver(proverConfig)
	if *gpus == "" {
		p.Run(*rerun)
		return
	}
	var ids []int
	for _, f := range strings.Split(*gpus, ",") {
		g, err := strconv.Atoi(strings.TrimSpace(f))
		if err != nil || g < 0 {
			panic("bad -gpus entry: " + f)
		}
		ids = append(ids, g)
	}
	if err := p.RunInProcess(*rerun, ids, *workers); err != nil {
		fmt.Println("prover failed:", err.Error())
		panic(err.Error())
	}
}
