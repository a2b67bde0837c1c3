#!/usr/bin/env python3
"""CLI with the reference's flags (main.py:13-22,46-105):

  python main.py --data_name synthetic_citeulike --model_choice mf --conf_choice best \
      --train_scheme neg_shared --eval_scheme whole@50 \
      --param_dict "{'reset_after_getconf': True, 'max_epoch': 5, 'loss': 'skip-gram', 'num_negatives': 0, \
                     'neg_loss_weight': 128, 'loss_gamma': 10, 'learn_rate': 0.5, 'neg_dist': 'unigram', \
                     'neg_sampling_power': 1, 'batch_size_p': 512, 'user_dim': 100, 'item_dim': 100}"
"""
import argparse
import ast
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--data_name", required=True)
    parser.add_argument("--model_choice", required=True)
    parser.add_argument("--conf_choice", required=True)
    parser.add_argument("--train_scheme", default="original")
    parser.add_argument("--eval_scheme", default="given")
    parser.add_argument("--param_dict", default=None)
    parser.add_argument("--pred_name", default=None)
    parser.add_argument("--gpu", default=None, type=str)
    parser.add_argument("--seed", default=4224, type=int, help="NumPy seed (the reference is unseeded)")
    args = parser.parse_args()
    # the reference's default is 'given' (main.py:18): pair AUC over sampled negatives through pandas / sklearn
    # (objectives.py:231-294), outside the measured path and not built -- refused HERE, before data, samplers
    # and a model are built and an epoch has run, not inside the first evaluation
    if not args.eval_scheme.startswith("whole"):
        parser.error("--eval_scheme %s is not built: use whole@K (e.g. whole@50, BASELINE's Recall@50)" % args.eval_scheme)
    if args.pred_name is not None:
        parser.error("--pred_name (pickling the (users x items) prediction matrices, objectives.py:317-318) is not built")
    param_dict = ast.literal_eval(args.param_dict) if args.param_dict is not None else None
    device = int(args.gpu) if args.gpu is not None else 0
    print("model_choice: %s \nconf_choice: %s" % (args.model_choice, args.conf_choice))

    import numpy as np
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model

    np.random.seed(args.seed)
    conf = get_conf(args.data_name, args.conf_choice, param_dict, model_choice=args.model_choice)
    eval_scheme = args.eval_scheme
    if eval_scheme.find("@") > 0:  # main.py:60-63
        p = eval_scheme.find("@")
        conf.eval_topk = int(eval_scheme[p + 1:])
        eval_scheme = eval_scheme[:p]
    if conf.optimizer == "sgd":
        print("[INFO] optimizer = sgd (sparse SGD scatter update, opt-in): the demo scripts' learn_rate 0.01 is an Adam "
              "step size -- plain SGD barely moves with it.")
    data_helper = get_data(args.data_name, conf, reverse_samping=True, rand_seed=12345, device=device)
    print(conf.__dict__)
    model_dict = get_model(conf, data_helper, args.model_choice, device=device)

    if args.train_scheme == "original":
        from nncf_amd.train_original import Trainer
    elif args.train_scheme == "neg_shared":
        from nncf_amd.train_neg_shared import Trainer
    elif args.train_scheme == "group_neg_shared":
        from nncf_amd.train_group_neg_shared import Trainer
    elif args.train_scheme == "reverse":
        from nncf_amd.train_reverse import Trainer
    elif args.train_scheme == "presample":
        from nncf_amd.train_presample import Trainer
    elif args.train_scheme == "group_sample":
        from nncf_amd.train_group_sample import Trainer
    elif args.train_scheme == "sampled_neg_shared":
        from nncf_amd.train_sampled_neg_shared import Trainer
    else:
        assert False, "[ERROR] Unknown train_scheme {}".format(args.train_scheme)
    trainer = Trainer(model_dict, conf, data_helper)
    trainer.train(eval_scheme)
    if args.pred_name is not None:
        trainer.predict(eval_scheme, args.pred_name)


if __name__ == "__main__":
    main()
