#!/bin/bash
# scripts/dev/build_variant.sh <name> "<extra hipcc flags>": a timing build of libpxsom.so under ark_analysis_amd/variants/
set -e
cd "$(dirname "$0")/../.."
PXSOM_HIPCC_EXTRA="$2" python -c "from ark_analysis_amd import _build; _build.build(force=True)"
cp ark_analysis_amd/libpxsom.so "ark_analysis_amd/variants/$1.so"
echo "built ark_analysis_amd/variants/$1.so with '$2'"
