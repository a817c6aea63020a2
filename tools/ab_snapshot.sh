#!/bin/bash
# Snapshot the current build into tools/ab/<name>/ (git-ignored; travels with gpurun) for same-box A/B runs:
#   tools/ab_snapshot.sh e1 && ... && gpurun -- 'for v in cur e1; do python tools/ab/$v/tools/ab_full.py; done'
set -e
name=${1:?name}
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/tools/ab/$name
rm -rf "$d"; mkdir -p "$d/motifscan_amd" "$d/tools" "$d/include"
cp "$root"/motifscan_amd/*.py "$root"/motifscan_amd/libmotifscan_amd.so "$d/motifscan_amd/"
cp -r "$root/motifscan_amd/data" "$d/motifscan_amd/"
cp "$root"/include/*.h "$d/include/"
cp "$root/tools/ab_full.py" "$root/tools/pf_account.py" "$d/tools/"
echo "snapshot $name: $(sha256sum "$root/motifscan_amd/csrc/ms_kernels.hip" | cut -c1-16)"
