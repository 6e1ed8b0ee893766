#!/bin/bash
# Every kernel of the library must be ONE function: a lambda that the inliner leaves as a function of its own keeps its captures
# (accumulators!) in memory -- the 7x7 instance of conv_mfma_kernel ran four times slower that way for most of round 3.  Compiles
# every .hip to assembly and fails on an outlined lambda (_ZZN...) or a call (s_swappc).   bash tools/check_device_calls.sh
# Also (round 4): tools/isa_hazard_check.py over the same assembly -- a transcendental result read by the very next vector
# instruction, which the compiler pads itself except inside an inline-asm statement.
here=$(cd "$(dirname "$0")" && pwd)
cd "$here/../synt_isic_amd/csrc" || exit 1
tmp=$(mktemp -d)
rc=0
for f in *.hip; do
    extra=""; [ "$f" = elementwise.hip ] && extra="-ffp-contract=off"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $extra -S --cuda-device-only "$f" -o "$tmp/${f%.hip}.s" 2>/dev/null || { echo "$f: does not compile"; rc=1; continue; }
    n=$(grep -c '^_ZZN' "$tmp/${f%.hip}.s"); c=$(grep -c 's_swappc' "$tmp/${f%.hip}.s")
    echo "$f: $n outlined lambdas, $c calls"
    [ "$n" -eq 0 ] && [ "$c" -eq 0 ] || rc=1
    python3 "$here/isa_hazard_check.py" "$tmp/${f%.hip}.s" || rc=1
done
rm -rf "$tmp"
exit $rc
