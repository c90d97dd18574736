cd /tmp && hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics $GRAFT_REPO_ROOT/scripts/ubench/lds_atomic_rate.hip -o lds_atomic_rate && ./lds_atomic_rate
