timeout 900 python -m pytest tests -m gpu -x -q -k "mask or abi" 2>&1 | tail -5
