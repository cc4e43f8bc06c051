#!/bin/bash
# Stage the read-only reference checkout into the git-ignored gpurun_scratch/reference so that ONE gpurun call can run the
# unmodified reference next to tensoir_amd.  Nothing staged is ever committed; `unstage` removes it.
# Usage: tools/stage_reference.sh [unstage]
set -eu
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
DST="$REPO/gpurun_scratch/reference"
if [ "${1:-}" = "unstage" ]; then rm -rf "$DST"; echo "removed $DST"; exit 0; fi
SRC="${TENSOIR_REFERENCE_SRC:-/root/reference}"
rm -rf "$DST"; mkdir -p "$DST"
(cd "$SRC" && tar cf - --exclude=__pycache__ --exclude=.git .) | (cd "$DST" && tar xf -)
git -C "$REPO" check-ignore -q "$DST" || { echo "refusing: $DST is not git-ignored"; rm -rf "$DST"; exit 1; }
du -sh "$DST"
