"""Driver: `python -m breakmer_amd.breakmer [options] <config file>` -- the reference's CLI
(breakmer.py:50-96): key=value config file (kmer_region.config:1-16) overlaid with the same options."""
from __future__ import annotations

import argparse
import logging
import os
import sys
import time


def parse_config_f(config_fn, opts):                                # breakmer.py:50-66
    cfg = {}
    with open(config_fn) as f:
        for line in f:
            line = line.strip()
            parts = line.split("=")
            if len(parts) == 1:
                print('Config line', line, ' not set correctly. Exiting.')
                sys.exit()
            k, v = parts
            cfg[k] = v
    cfg.update(vars(opts))
    return cfg


def build_parser():                                                 # breakmer.py:70-86
    p = argparse.ArgumentParser(usage='%(prog)s [options] <config file name>',
                                description="Script to identify structural variants within targeted locations.")
    p.add_argument('config')
    p.add_argument('-l', '--log_level', dest='log_level', default='DEBUG')
    p.add_argument('-a', '--keep_repeat_regions', dest='keep_repeat_regions', default=False, action='store_true')
    p.add_argument('-p', '--preset_ref_data', dest='preset_ref_data', default=False, action='store_true')
    p.add_argument('-s', '--indel_size', dest='indel_size', default=15, type=int)
    p.add_argument('-c', '--trl_sr_thresh', dest='trl_sr_thresh', default=2, type=int)
    p.add_argument('-d', '--indel_sr_thresh', dest='indel_sr_thresh', default=5, type=int)
    p.add_argument('-r', '--rearr_sr_thresh', dest='rearr_sr_thresh', default=3, type=int)
    p.add_argument('-g', '--gene_list', dest='gene_list', default=None)
    p.add_argument('-k', '--keep_intron_vars', dest='keep_intron_vars', default=False, action='store_true')
    p.add_argument('-v', '--var_filter', dest='var_filter', default='all')
    p.add_argument('-m', '--rearr_min_seg_len', dest='rearr_minseg_len', default=30, type=int)
    p.add_argument('-n', '--trl_min_seg_len', dest='trl_minseg_len', default=25, type=int)
    p.add_argument('-t', '--align_thresh', dest='align_thresh', default=.90, type=float)   # parsed, never read (as in the reference)
    p.add_argument('-z', '--no_output_header', dest='no_output_header', default=False, action='store_true')
    return p


def setup_logger(config_d, name='root'):                            # utils.py:98-117
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    os.makedirs(config_d['analysis_dir'], exist_ok=True)
    fh = logging.FileHandler(os.path.join(config_d['analysis_dir'], 'log.txt'), mode='w')
    fh.setLevel(logging.DEBUG)
    ch = logging.StreamHandler()
    ch.setLevel(logging.ERROR)
    fmt = logging.Formatter(fmt='%(asctime)s - %(name)s - %(levelname)s - %(message)s', datefmt='%m/%d/%Y %I:%M:%S %p')
    fh.setFormatter(fmt)
    ch.setFormatter(fmt)
    logger.addHandler(fh)
    logger.addHandler(ch)


def main(argv=None):
    from .sv_processor import runner
    args = build_parser().parse_args(argv)
    config_fn = args.config
    del args.config
    tic = time.perf_counter()
    config_d = parse_config_f(config_fn, args)
    setup_logger(config_d, 'root')
    rank, world, collate, status_exchange = 0, 1, None, None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:                   # one process per GPU, regions sharded by rank
        import torch
        import torch.distributed as td
        from .collate import collate_results, exchange_status
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        td.init_process_group("nccl", device_id=torch.device("cuda", local))
        rank, world, collate, status_exchange = td.get_rank(), td.get_world_size(), collate_results, exchange_status
    r = runner(config_d, rank=rank, world=world, collate=collate, status_exchange=status_exchange)
    r.run(tic)
    logging.getLogger('root').info('Analysis complete, %s' % str(time.perf_counter() - tic))
    return 3 if r.failed_targets else 0          # 3: the run is complete but some targets have no result (log: which and why)


if __name__ == '__main__':
    import sys
    sys.exit(main())
