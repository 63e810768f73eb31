package zkporgpu

// The Poseidon half of the boundary: account leaf hashes, the fixed-depth account tree as an object resident in HBM, the dense
// one-shot tree build and batched proof verification.  Replaces, behind the same method set, what the reference reaches through
//   utils.AccountInfoToHash / ComputeUserAssetsCommitment   src/utils/utils.go:744-750, 188-221
//   merkletree.FixedDepthMerkleTree Set/Build/Root/Get/GetProof   src/utils/merkletree/merkletree.go:179-308
//   utils.NewAccountTree / VerifyMerkleProof                src/utils/account_tree.go:14-29
//   buildAccountTree                                        src/witness/main.go:130-199
// This package cannot import src/utils (utils.NewAccountTree calls into it — go/witness.patch), so accounts cross as the flat
// mirrors below; the conversion from utils.AccountInfo is three lines in the patch.  NOT COMPILED in the authoring image.

/*
#include <stdlib.h>
#include "zkpor.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"math/big"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

// Asset mirrors utils.AccountAsset (src/utils/types.go:25-32) = zkpor_asset_t (48 bytes).
type Asset struct {
	Equity, Debt, Loan, Margin, PortfolioMargin uint64
	Index, _pad                                 uint32
}

// Account mirrors utils.AccountInfo (src/utils/types.go:34-41) = zkpor_account_t (88 bytes): totals as 128-bit little-endian words.
type Account struct {
	ID                       [32]byte // AccountId, big-endian
	Equity, Debt, Collateral [2]uint64
	NAssets, AssetOff        uint32
}

// Put128 stores a non-negative big.Int below 2^128 (TotalEquity / TotalDebt / TotalCollateral) into one of the Account fields.
func Put128(dst *[2]uint64, v *big.Int) error {
	if v.Sign() < 0 || v.BitLen() > 128 {
		return fmt.Errorf("zkporgpu: account total %s does not fit 128 bits", v.String())
	}
	var be [16]byte
	v.FillBytes(be[:])
	dst[1] = uint64(be[0])<<56 | uint64(be[1])<<48 | uint64(be[2])<<40 | uint64(be[3])<<32 | uint64(be[4])<<24 | uint64(be[5])<<16 | uint64(be[6])<<8 | uint64(be[7])
	dst[0] = uint64(be[8])<<56 | uint64(be[9])<<48 | uint64(be[10])<<40 | uint64(be[11])<<32 | uint64(be[12])<<24 | uint64(be[13])<<16 | uint64(be[14])<<8 | uint64(be[15])
	return nil
}

func accPtr(a []Account) *C.zkpor_account_t {
	if len(a) == 0 {
		return nil
	}
	return (*C.zkpor_account_t)(unsafe.Pointer(&a[0]))
}
func assetPtr(a []Asset) *C.zkpor_asset_t {
	if len(a) == 0 {
		return nil
	}
	return (*C.zkpor_asset_t)(unsafe.Pointer(&a[0]))
}
func bytePtr(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// AccountLeaves = utils.AccountInfoToHash for every account (assets padded to `tier` exactly as PaddingAccountAssets does):
// n x 32 bytes, big-endian, in account order.
func (c *Context) AccountLeaves(accounts []Account, assets []Asset, tier int) ([]byte, error) {
	out := make([]byte, 32*len(accounts)+1)
	e := c.err(C.zkpor_poseidon_leaves(c.h, accPtr(accounts), assetPtr(assets), C.size_t(len(assets)), C.size_t(len(accounts)), C.int(tier), bytePtr(out)))
	return out[:32*len(accounts)], e
}

// PoseidonHash = poseidon.Poseidon(inputs...) for `count` independent inputs of `length` elements each.
func (c *Context) PoseidonHash(inputs []fr.Element, length, count int) ([]fr.Element, error) {
	if length*count != len(inputs) || count == 0 {
		return nil, errors.New("zkporgpu: PoseidonHash: len(inputs) != length * count")
	}
	out := make([]fr.Element, count)
	e := c.err(C.zkpor_poseidon_hash(c.h, (*C.uint64_t)(unsafe.Pointer(&inputs[0])), C.size_t(length), C.size_t(count), (*C.uint64_t)(unsafe.Pointer(&out[0]))))
	return out, e
}

// MerkleBuild = NewFixedDepthMerkleTree + Set(0..n-1) + Build + Root in one call for a dense prefix of leaves (32 B big-endian each).
func (c *Context) MerkleBuild(leaves []byte, depth int, nilLeaf []byte) (root []byte, err error) {
	if len(leaves)%32 != 0 || len(nilLeaf) != 32 {
		return nil, errors.New("zkporgpu: MerkleBuild: leaves and nilLeaf are 32-byte hashes")
	}
	root = make([]byte, 32)
	err = c.err(C.zkpor_merkle_build(c.h, bytePtr(leaves), C.size_t(len(leaves)/32), C.int(depth), bytePtr(nilLeaf), nil, bytePtr(root)))
	return
}

// AccountTree has the method set the reference uses of *merkletree.FixedDepthMerkleTree (Set, Build, Root, Get, GetProof), so that
// utils.NewAccountTree can hand it out behind the utils.AccountTree interface of go/witness.patch; plus the bulk forms a GPU wants.
type AccountTree struct {
	c     *Context
	h     *C.zkpor_tree
	depth int
}

// NewAccountTree = merkletree.NewFixedDepthMerkleTree(depth, nilLeaf, poseidon.NewPoseidon, capacity) (merkletree.go:137-176).
func (c *Context) NewAccountTree(depth int, nilLeaf []byte, capacity int) (*AccountTree, error) {
	if len(nilLeaf) != 32 {
		return nil, errors.New("zkporgpu: nil leaf hash must be 32 bytes")
	}
	t := &AccountTree{c: c, depth: depth}
	if e := c.err(C.zkpor_tree_create(c.h, C.int(depth), bytePtr(nilLeaf), C.uint64_t(capacity), &t.h)); e != nil {
		return nil, e
	}
	return t, nil
}

func (t *AccountTree) Close() {
	if t.h != nil {
		C.zkpor_tree_destroy(t.h)
		t.h = nil
	}
}

// Set (merkletree.go:179-187).  One PCIe round trip per leaf: fine for tests, use SetMany / SetAccounts for data sets.
func (t *AccountTree) Set(key uint32, value []byte) error {
	return t.SetMany([]uint32{key}, value)
}

// SetMany stores len(keys) leaves (values: 32 bytes each, concatenated); a key beyond the capacity fails the whole call.
func (t *AccountTree) SetMany(keys []uint32, values []byte) error {
	if len(values) != 32*len(keys) {
		return errors.New("zkporgpu: SetMany: values must hold 32 bytes per key")
	}
	if len(keys) == 0 {
		return nil
	}
	return t.c.err(C.zkpor_tree_set(t.h, u32ptr(keys), bytePtr(values), C.size_t(len(keys))))
}

// SetAccounts is buildAccountTree's inner loop (src/witness/main.go:175-185) for one chunk of accounts with CONTIGUOUS indices
// firstKey .. firstKey+len-1: leaf hashes are computed on the device and stored without crossing PCIe.
func (t *AccountTree) SetAccounts(firstKey uint32, accounts []Account, assets []Asset, tier int) error {
	if len(accounts) == 0 {
		return nil
	}
	return t.c.err(C.zkpor_tree_set_accounts(t.h, C.uint64_t(firstKey), accPtr(accounts), assetPtr(assets), C.size_t(len(assets)),
		C.size_t(len(accounts)), C.int(tier), nil, 0, nil))
}

// Build (merkletree.go:192-279): every internal node above a leaf set since the last Build.
func (t *AccountTree) Build() {
	if e := t.c.err(C.zkpor_tree_build(t.h)); e != nil {
		panic("zkporgpu: tree build: " + e.Error()) // the reference's Build has no error path either
	}
}

// Root (merkletree.go:282-284).
func (t *AccountTree) Root() []byte {
	out := make([]byte, 32)
	if e := t.c.err(C.zkpor_tree_root(t.h, bytePtr(out))); e != nil {
		panic("zkporgpu: tree root: " + e.Error())
	}
	return out
}

// Get (merkletree.go:287-294).
func (t *AccountTree) Get(key uint32) []byte {
	out := make([]byte, 32)
	if e := t.c.err(C.zkpor_tree_get(t.h, &[]C.uint32_t{C.uint32_t(key)}[0], 1, bytePtr(out))); e != nil {
		panic("zkporgpu: tree get: " + e.Error())
	}
	return out
}

// GetProof (merkletree.go:297-308): depth siblings, leaf level first.
func (t *AccountTree) GetProof(key uint32) ([][]byte, error) {
	p, err := t.GetProofs([]uint32{key})
	if err != nil {
		return nil, err
	}
	return p[0], nil
}

// GetProofs fetches the proofs of a whole batch of users in one call (fillCreateUserOp runs U of them per batch, witness.go:323).
func (t *AccountTree) GetProofs(keys []uint32) ([][][]byte, error) {
	if len(keys) == 0 {
		return nil, nil
	}
	flat := make([]byte, len(keys)*t.depth*32)
	if e := t.c.err(C.zkpor_tree_get_proofs(t.h, u32ptr(keys), C.size_t(len(keys)), bytePtr(flat))); e != nil {
		return nil, e
	}
	out := make([][][]byte, len(keys))
	for i := range keys {
		out[i] = make([][]byte, t.depth)
		for l := 0; l < t.depth; l++ {
			off := (i*t.depth + l) * 32
			out[i][l] = flat[off : off+32 : off+32]
		}
	}
	return out, nil
}

// VerifyMerkleProofs = merkletree.VerifyProof (merkletree.go:334-355) for n (key, leaf, proof) triples against one root.
func (c *Context) VerifyMerkleProofs(root []byte, keys []uint32, proofs [][][]byte, leaves [][]byte, depth int) ([]bool, error) {
	n := len(keys)
	if len(proofs) != n || len(leaves) != n || len(root) != 32 {
		return nil, errors.New("zkporgpu: VerifyMerkleProofs: one proof and one leaf per key")
	}
	flatP := make([]byte, 0, n*depth*32)
	flatL := make([]byte, 0, n*32)
	for i := 0; i < n; i++ {
		if len(proofs[i]) != depth || len(leaves[i]) != 32 {
			return nil, errors.New("zkporgpu: VerifyMerkleProofs: malformed proof")
		}
		for _, s := range proofs[i] {
			flatP = append(flatP, s...)
		}
		flatL = append(flatL, leaves[i]...)
	}
	ok := make([]byte, n+1)
	if e := c.err(C.zkpor_merkle_verify_proofs(c.h, bytePtr(root), u32ptr(keys), bytePtr(flatP), bytePtr(flatL), C.size_t(n), C.int(depth), bytePtr(ok))); e != nil {
		return nil, e
	}
	res := make([]bool, n)
	for i := range res {
		res[i] = ok[i] != 0
	}
	return res, nil
}
