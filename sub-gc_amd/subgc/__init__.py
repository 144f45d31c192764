"""subgc: MI355X-native Sub-GC hot path (GCN message passing + sGPN + attention-LSTM decoder).

    import subgc.models as models; model = models.setup(opt).cuda()

Compute goes through the C ABI of libsubgc_hip.so (include/subgc_hip.h); importing this package
does not need a GPU, running the model does.
"""
__version__ = "0.1.0"
