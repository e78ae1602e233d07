"""CPU oracle — TEST INFRASTRUCTURE ONLY (see gp_oracle.py header).  Never imported by the product."""
