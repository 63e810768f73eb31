package zkporgpu

/*
#include "zkpor.h"
*/
import "C"

import "fmt"

// abiVersion is the value of ZKPOR_ABI_VERSION this package was written against (include/zkpor.h).  The header's own value is
// compiled in through cgo; the LIBRARY's value is asked at run time: a libzkpor.so older or newer than the header would take
// arguments in other positions (z_order joined zkpor_pk_load_gnark* in the middle of the list in version 3; version 4: the sharded computeH's steps under the six-transform schedule) and nothing else
// would notice.  NOT COMPILED in the authoring image — go/README.md.
const abiVersion = 4

func init() {
	if C.ZKPOR_ABI_VERSION != abiVersion {
		panic(fmt.Sprintf("zkporgpu: include/zkpor.h declares ABI version %d, this package was written against %d", int(C.ZKPOR_ABI_VERSION), abiVersion))
	}
	if got := int(C.zkpor_abi_version()); got != abiVersion {
		panic(fmt.Sprintf("zkporgpu: libzkpor.so speaks ABI version %d, this package was written against %d", got, abiVersion))
	}
}
