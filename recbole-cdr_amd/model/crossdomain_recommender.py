"""Abstract cross-domain recommender = the drop-in boundary (recbole_cdr/model/crossdomain_recommender.py:14-51).

Same constructor contract ``(config, dataset)``, same attribute names, same ``set_phase`` hook; the third-party
``recbole.model.abstract_recommender.AbstractRecommender`` surface the trainer relies on (``calculate_loss`` /
``predict`` / ``full_sort_predict`` / ``other_parameter`` / ``load_other_parameter``) is restated here because recbole
is not a dependency of this package.
"""
import numpy as np
import torch.nn as nn
from torch.nn.init import xavier_normal_, constant_

from ..utils import ModelType


def xavier_normal_initialization(module):
    """recbole.model.init.xavier_normal_initialization: Embedding / Linear weights ~ xavier normal, biases 0."""
    if isinstance(module, nn.Embedding):
        xavier_normal_(module.weight.data)
    elif isinstance(module, nn.Linear):
        xavier_normal_(module.weight.data)
        if module.bias is not None:
            constant_(module.bias.data, 0)


class CrossDomainRecommender(nn.Module):
    type = ModelType.CROSSDOMAIN

    def __init__(self, config, dataset):
        super().__init__()
        # Per domain d in {source, target}: <D>_USER_ID / <D>_ITEM_ID / <D>_NEG_ITEM_ID field names and
        # <d>_num_users / <d>_num_items (overlap + that domain's own ids; the tables themselves are union-sized).
        for dom in ('source', 'target'):
            single = getattr(dataset, dom + '_domain_dataset')
            tag = dom.upper()
            uid, iid = single.uid_field, single.iid_field
            setattr(self, tag + '_USER_ID', uid)
            setattr(self, tag + '_ITEM_ID', iid)
            setattr(self, tag + '_NEG_ITEM_ID', config[dom + '_domain']['NEG_PREFIX'] + iid)
            setattr(self, dom + '_num_users', single.num(uid))
            setattr(self, dom + '_num_items', single.num(iid))
        # Union id space: [0] = PAD, [1, overlapped) shared, then target-only, then source-only ids.
        self.total_num_users, self.total_num_items = dataset.num_total_user, dataset.num_total_item
        self.overlapped_num_users, self.overlapped_num_items = dataset.num_overlap_user, dataset.num_overlap_item
        self.OVERLAP_ID = dataset.overlap_id_field
        self.device = config['device']

    def one_sided_overlap_mode(self):
        """'overlap_users' | 'overlap_items' | 'non_overlap' for the models that need the domains to share EITHER users OR items
        (emcdr.py:33-40 and the same guard in conet / sscdr / natr / deepapf / dcdcsr).  An overlap count of 1 is the PAD id alone."""
        nu, ni = self.overlapped_num_users, self.overlapped_num_items
        if nu > 1 and ni > 1:
            raise AssertionError(f'{type(self).__name__} handles a user-overlapped or an item-overlapped pair of domains, not both '
                                 f'({nu - 1} shared users and {ni - 1} shared items in this dataset)')
        return 'overlap_users' if nu > 1 else 'overlap_items' if ni > 1 else 'non_overlap'

    def set_phase(self, phase):
        pass

    def graph_key(self):
        """Hashable tag of what ``calculate_loss`` would enqueue right now, or None when the step must not be captured in a hipGraph
        (``Trainer`` replays one captured step per batch shape and key: graph_step.GraphedTrainStep).  A model answers with a key
        only when its loss does ALL per-step work on the device -- a host draw, a host counter or a tensor rebuilt per phase visit
        would be frozen into the capture.  Default: not capturable (the eager loop)."""
        return None

    def on_train_steps(self):
        """Called by ``Trainer`` after training steps that ran WITHOUT the model's Python (hipGraph replays): the place to drop
        anything ``calculate_loss`` would have invalidated on the host -- e.g. BiTGCF's cached propagated embeddings
        (bitgcf.py:146-148 of the reference clears them at the top of every ``calculate_loss``).  Default: nothing cached."""

    def calculate_loss(self, interaction):
        raise NotImplementedError

    def predict(self, interaction):
        raise NotImplementedError

    def full_sort_predict(self, interaction):
        raise NotImplementedError

    def other_parameter(self):
        if hasattr(self, 'other_parameter_name'):
            return {key: getattr(self, key) for key in self.other_parameter_name}
        return dict()

    def load_other_parameter(self, para):
        if para is None:
            return
        for key, value in para.items():
            setattr(self, key, value)

    def __str__(self):
        params = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        return super().__str__() + f'\nTrainable parameters: {params}'
