"""Host-side utilities of the MI355X build (config composition, sweep farm)."""
