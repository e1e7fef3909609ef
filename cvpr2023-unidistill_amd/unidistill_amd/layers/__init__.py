"""Host-side mirrors of the reference's layer modules on the hot path (same constructor
arguments, forward signatures and state_dict key names; own implementation on the HIP ops)."""
