"""Host-side mirror of the reference package `models/STSwinNet_SNN` (same class names, constructor
signatures and state_dict key layout), executing on the HIP engine (`sdformerflow_amd.engine`)."""
