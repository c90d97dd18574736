"""Mirrors of the reference's Pixie pipeline modules (same module / function / class names and
signatures as ``ark.phenotyping.*``), with the two pyFlowSOM calls and the per-cluster reduction
running on MI355X through libpxsom.so.  See INTEGRATION.md."""
