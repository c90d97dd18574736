time python bench.py
