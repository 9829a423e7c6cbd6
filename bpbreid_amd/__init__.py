"""bpbreid_amd: the BPBReID hot path (backbone, part-attention head, GiLt loss, Adam, part-based ranking) on MI355X."""
