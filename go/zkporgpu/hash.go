package zkporgpu

import (
	"hash"

	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

// hashToField mirrors the default gnark installs in backend.NewProverConfig (hash_to_field over SHA-256 with the commitment
// domain-separation tag): it is only reached if a caller passes a config without one.
type h2f struct {
	dst  []byte
	data []byte
}

func hashToField(dst []byte) hash.Hash { return &h2f{dst: dst} }

func (h *h2f) Write(p []byte) (int, error) { h.data = append(h.data, p...); return len(p), nil }
func (h *h2f) Sum(b []byte) []byte {
	res, err := fr.Hash(h.data, h.dst, 1)
	if err != nil {
		panic(err)
	}
	bts := res[0].Bytes()
	return append(b, bts[:]...)
}
func (h *h2f) Reset()         { h.data = h.data[:0] }
func (h *h2f) Size() int      { return fr.Bytes }
func (h *h2f) BlockSize() int { return 64 }
