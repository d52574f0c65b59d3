#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
#
# Builds the *real* walexi/gnn.cpp reference (CPU path) into oracle/_ref/ref_driver so that the
# CPU restatement in oracle/gcn_oracle.c and the golden fixtures under tests/golden/ can be pinned
# against outputs of the reference itself.
#
# The reference HEAD does not compile as shipped (typo'd include, duplicate declaration, two
# definitions without declarations, ...).  None of the blockers touches arithmetic.  This recipe
# therefore makes a *throw-away* patched view in a mktemp dir, compiles from it, and deletes it:
# no reference source is ever copied into this repository; only the binary lands in oracle/_ref/
# (git-ignored).  If /root/reference is absent (GPU box) the script is a no-op.
set -euo pipefail
REF=${GNNX_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/include" ]; then
  echo "build_ref: $REF not present; skipping (prebuilt oracle/_ref is used if it travelled)"; exit 0
fi
mkdir -p "$OUT"
if [ "$OUT/ref_driver" -nt "$HERE/ref_driver.cpp" ] && [ "$OUT/ref_driver" -nt "$HERE/build_ref.sh" ]; then
  echo "build_ref: up to date"; exit 0
fi
T="$(mktemp -d /tmp/gnnx_refbuild.XXXXXX)"
trap 'rm -rf "$T"' EXIT
cp -r "$REF/include" "$REF/src" "$T/"
chmod -R u+w "$T"
cd "$T"
# (1) operation.h:9 / nn.cpp:6 include "util.h"; the file is utils.h.
sed -i 's/#include "util.h"/#include "utils.h"/' include/operation.h src/nn.cpp
# (2) graph.h:132 uses tensor<int>/tptr<float> unqualified outside namespace cyg.
sed -i '132s/const tensor<int>/const cyg::tensor<int>/; 132s/const tptr<float> \&x/const cyg::tptr<float> \&x/; 132s/const tptr<float>\* others/const cyg::tptr<float>* others/' include/graph.h
# (3) graph.cpp:155-156 declares `auto out` twice in the (unused) base propagate.
sed -i '155s/auto out = /auto out0 = /' src/graph.cpp
# (4) nn.cpp:455-461 defines nn::Embedding which is declared nowhere.
sed -i '455,461d' src/nn.cpp
# (5) tensor.h calls generate_random<T>(...) but only a non-template is declared (utils.h:94).
sed -i '94a template <class T> T generate_random(const float \&low, const float \&high) { return static_cast<T>(generate_random(low, high)); }' include/utils.h
# (6) tensor.cpp:15-24 defines a cyg::randn overload with no declaration.
sed -i '15,24d' src/tensor.cpp
# (7) nn.cpp defines tanh/sigmoid/cross_entropy_loss `inline` but nn.h declares them non-inline.
sed -i 's/^inline cyg::tptr<float> nn::/cyg::tptr<float> nn::/' src/nn.cpp
g++ -std=c++20 -O2 -fpermissive -w \
    -include memory -include climits -include ctime -include stdexcept -include tuple -include string \
    -Iinclude "$HERE/ref_driver.cpp" src/utils.cpp src/tensor.cpp src/nn.cpp src/graph.cpp \
    -o "$OUT/ref_driver"
echo "build_ref: built $OUT/ref_driver"
