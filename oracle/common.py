"""Id-space contract shared by every oracle model (crossdomain_recommender.py:21-48, dataset.py:384-399).

ids: [0]=PAD, [1,OU) overlapped, [OU,OU+TO) target-only, [OU+TO,total) source-only; tables are allocated at
the UNION sizes (emcdr.py:67-71).  ``OU``/``OI`` already include the PAD row.
"""
from dataclasses import dataclass


@dataclass
class IdSpace:
    OU: int
    TOU: int
    SOU: int
    OI: int
    TOI: int
    SOI: int

    @property
    def total_num_users(self):
        return self.OU + self.TOU + self.SOU

    @property
    def total_num_items(self):
        return self.OI + self.TOI + self.SOI

    @property
    def target_num_users(self):
        return self.OU + self.TOU

    @property
    def target_num_items(self):
        return self.OI + self.TOI

    @property
    def source_num_users(self):
        return self.OU + self.SOU

    @property
    def source_num_items(self):
        return self.OI + self.SOI

    @property
    def overlapped_num_users(self):
        return self.OU

    @property
    def overlapped_num_items(self):
        return self.OI

    @property
    def mode(self):
        # emcdr.py:33-40 / conet.py:39-46 / sscdr.py:32-39
        if self.OU > 1:
            return 'overlap_users'
        if self.OI > 1:
            return 'overlap_items'
        return 'non_overlap'
