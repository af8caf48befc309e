#!/bin/bash
# Builds and tests the cgo binding of libkmcpgpu.so inside a checkout of shenwei356/kmcp (v0.9.5) — one command for a maintainer
# with a Go toolchain, ROCm and an MI300/MI355-class GPU:
#
#     shim/build.sh /path/to/kmcp-source [device]
#
# What it does:  1. builds libkmcpgpu.so (make in kmcp_amd/csrc; needs hipcc) unless it is there already;
#                2. copies shim/kmcp_gpu.go + shim/kmcp_gpu_test.go into <kmcp>/kmcp/cmd/ and include/kmcp_gpu.h into <kmcp>/include/
#                   (the layout the `#cgo CFLAGS: -I${SRCDIR}/../../include` line of kmcp_gpu.go assumes);
#                3. runs `go vet` and `go test -tags kmcpgpu -run TestGPUSearchFixture ./kmcp/cmd/` against shim/testdata (expected.tsv there was
#                   printed by this repository's CPU oracle — oracle/kmcp_oracle.c, the restatement of the reference's algorithm pinned
#                   by the reference's demo tables — NOT by the Go reference, which could not be built where the fixture was made:
#                   tests/golden/make_shim_fixture.py; run `kmcp search` on shim/testdata yourself to close that loop.  The same
#                   fixture is checked against kmcp-search on the GPU by this repository's own tests);
#                4. builds the kmcp binary with the tag (the wiring of NewGPUSearchEngine into search.go:400 is shown in
#                   kmcp_gpu.go and INTEGRATION.md; without it the binary simply carries the binding).
# There is no Go toolchain in the image this repository was built in: this script and the two .go files have never been run there.
# The C call sequence of the binding is what tests/shim_replay.c replays on the GPU with every test run.
set -eu
[ $# -ge 1 ] || { echo "usage: $0 /path/to/kmcp-source [device]" >&2; exit 2; }
KMCP_SRC=$(cd "$1" && pwd)
DEVICE=${2:-0}
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(dirname "$HERE")
[ -f "$KMCP_SRC/kmcp/cmd/util-db-search.go" ] || { echo "$KMCP_SRC does not look like a checkout of shenwei356/kmcp (kmcp/cmd/util-db-search.go missing)" >&2; exit 2; }
command -v go > /dev/null || { echo "no Go toolchain on PATH" >&2; exit 2; }
if [ ! -f "$REPO/kmcp_amd/libkmcpgpu.so" ]; then
  make -j8 -C "$REPO/kmcp_amd/csrc" ARCH="${ARCH:-gfx950}"
fi
mkdir -p "$KMCP_SRC/include"
cp "$REPO/include/kmcp_gpu.h" "$KMCP_SRC/include/"
cp "$HERE/kmcp_gpu.go" "$HERE/kmcp_gpu_test.go" "$KMCP_SRC/kmcp/cmd/"
cd "$KMCP_SRC"
export CGO_ENABLED=1
export CGO_LDFLAGS="-L$REPO/kmcp_amd -Wl,-rpath,$REPO/kmcp_amd"
export KMCP_GPU_FIXTURE="$HERE/testdata"
export HIP_VISIBLE_DEVICES="$DEVICE"
go vet -tags kmcpgpu ./kmcp/cmd/
go test -tags kmcpgpu -run TestGPUSearchFixture -count=1 -v ./kmcp/cmd/
go build -tags kmcpgpu -o "$REPO/kmcp_amd/kmcp-go" ./kmcp
echo "ok: binding tested against shim/testdata; binary with the binding: $REPO/kmcp_amd/kmcp-go"
