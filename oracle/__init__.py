"""CPU oracle for the IPK scoring hot path -- TEST INFRASTRUCTURE ONLY (see ipk_oracle.c header)."""
