"""Test-only CPU oracle (see lb_oracle.py header). Never imported by lagrangebench_amd/."""
