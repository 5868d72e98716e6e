"""Host-side mirror of the reference package `models/STSwinNet` for the pieces on the hot path:
window partition / reverse / mask and the ANN cosine WindowAttention3D (swin_transformer3D_v2.py)."""
