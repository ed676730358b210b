#!/bin/bash
# Container-only: regenerate tests/golden/*.rtxb from the reference's own host code.
# Nothing under /root/reference is copied; its headers and SceneManager.cpp are compiled where they lie.
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/gen_golden_blocks
g++ -std=c++11 -O1 -ffp-contract=off -include cfloat \
    -I"$REF/src" -I"$REF/common" -I"$REF/external_sources/glm" -I"$REF/external_sources/glad/include" \
    -I"$HERE/raytracing_opengl_amd/csrc/host" -DASSETS_DIR='""' \
    "$HERE/tools/gen_golden_blocks.cpp" "$REF/src/SceneManager.cpp" \
    -Wl,--unresolved-symbols=ignore-all -o "$OUT"
"$OUT" "$HERE/tests/golden"
( cd "$HERE/tests/golden" && sha256sum *.rtxb *.npz jpeg/* > SHA256SUMS )
