// Module of the pin harness: it imports the UPSTREAM packages, not this repository's shim.
//
//   cd bindings/go/pin && go test -tags pin -run TestPin -v
//
// With a local checkout of shenwei356/bio instead of the module proxy:
//   go mod edit -replace github.com/shenwei356/bio=/path/to/bio && go mod tidy
module biosketch/pin

go 1.22

require (
	github.com/shenwei356/bio v0.13.8
	github.com/zeebo/wyhash v0.0.1
)
