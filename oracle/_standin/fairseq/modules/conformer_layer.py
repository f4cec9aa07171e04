class ConformerWav2Vec2EncoderLayer:  # conformer branch only; never built by shipped configs
    def __init__(self, *a, **k):
        raise NotImplementedError
