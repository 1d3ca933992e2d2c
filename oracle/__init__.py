"""CPU oracle for the ky path-tracing hot path: TEST INFRASTRUCTURE ONLY (see ky_oracle.cpp header)."""
