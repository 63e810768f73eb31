// Stand-alone module file for building/testing the package outside the reference tree.  Inside the reference the package is
// simply copied to src/prover/zkporgpu and uses the reference's own go.mod (same requirements, same replace directives).
module github.com/binance/zkmerkle-proof-of-solvency/src/prover/zkporgpu

go 1.22

require (
	github.com/consensys/gnark v0.10.0
	github.com/consensys/gnark-crypto v0.14.0
)

// the forks the reference pins (/root/reference/go.mod:57-60)
replace (
	github.com/consensys/gnark => github.com/bnb-chain/gnark v0.10.1-0.20240910145009-4b5261061f04
	github.com/consensys/gnark-crypto => github.com/bnb-chain/gnark-crypto v0.14.1-0.20240910145340-609ab3a7eb9b
)
