"""TEST INFRASTRUCTURE ONLY: CPU checker for the HIP path (see ilqr_oracle.h)."""
