from dataclasses import dataclass


@dataclass
class FairseqDataclass:
    pass
