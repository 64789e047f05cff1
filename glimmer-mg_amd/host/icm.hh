//  icm.hh -- MI355X-native drop-in for the reference's Interpolated Context
//  Model classes (scoring: ICM_t; training: ICM_Training_t).
//
//  Same public interface, constants, field names and error behaviour as the
//  reference's  ICM_t  (src/ICM/icm.hh:26-84,106-180,303), so that code written
//  against it (src/Glimmer/glimmer3.cc, glimmer-mg.cc, glimmer_base.cc,
//  src/ICM/build-icm.cc) recompiles unchanged.  (src/ICM/score-fixed.cc does NOT: it also
//  needs Fixed_Length_ICM_t, src/ICM/icm.hh:216-289, which is out of scope -- SURVEY 2, row 10.)  Everything that computes a
//  score goes through the extern "C" HIP layer declared in include/gmg.h;
//  there is no CPU scoring path in this class.  Model I/O and the null-model
//  builder are host code, as in the reference.
//
//  Additions (not in the reference) are grouped at the end of the class:
//  batch entry points and access to the device mirror.

#ifndef GMG_HOST_ICM_HH_INCLUDED
#define GMG_HOST_ICM_HH_INCLUDED

#include <stdio.h>
#include <string>
#include <vector>

struct gmg_model;   // include/gmg.h

// ---- constants of src/ICM/icm.hh:21-80 (callers and the training code use them) ----
#define  STORE_MUT_INFO  1
const int  ALPHABET_SIZE = 4;
const int  ALPHA_SQUARED = (ALPHABET_SIZE * ALPHABET_SIZE);
const char  ALPHA_STRING [] = "acgt";
const int  DEFAULT_MODEL_LEN = 12;
const int  ID_STRING_LEN = 150;
const int  DEFAULT_MODEL_DEPTH = 7;
const unsigned  NUM_FIXED_LENGTH_PARAMS = 6;
const int  DEFAULT_PERIODICITY = 3;
const char  MAX_MI_CHAR = '*';
const char  SEPARATOR_CHAR = '|';
const int  ICM_VERSION_ID = 200;
//  used by the training side and by glimmer3's Integerize_Scores (glimmer3.cc:627)
const int  NUM_CHI2_ENTRIES = 7;
const float  CHI2_VAL [NUM_CHI2_ENTRIES] = {2.37, 4.11, 6.25, 7.81, 9.35, 11.3, 12.8};
const float  CHI2_SIGNIFICANCE [NUM_CHI2_ENTRIES] = {0.50, 0.75, 0.90, 0.95, 0.975, 0.99, 0.995};
const double  MUT_INFO_BIAS = 0.03;
const double  MAX_LOG_DIFF = -46.0;
const double  MUT_INFO_EPSILON = 1e-4;
const double  PSEUDO_COUNT = 0.001;
const int  SAMPLE_SIZE_BOUND = 400;

#define  PARENT(x) ((int) ((x) - 1) / ALPHABET_SIZE)


//  src/ICM/icm.hh:106-113 (STORE_MUT_INFO = 1 layout, 24 bytes)
struct  ICM_Score_Node_t
  {
   short int  mut_info_pos;
   float  mut_info;
   float  prob [ALPHABET_SIZE];
  };


class  ICM_t
  {
  protected:
   bool  empty;
   int  model_len;
   int  model_depth;
   int  periodicity;
   int  num_nodes;
   ICM_Score_Node_t  * * score;

  public:
   ICM_t
       (int m = DEFAULT_MODEL_LEN,
        int d = DEFAULT_MODEL_DEPTH,
        int p = DEFAULT_PERIODICITY);
   ~ ICM_t
       ();

   int  Get_Model_Len
       (void)
     { return  model_len; }
   int  Get_Periodicity
       (void)
     { return  periodicity; }

   void  Build_Indep_WO_Stops
       (double gc_frac, const std::vector <const char *> & stop_codon);
   void  Build_Reverse_Codon_WO_Stops
       (double codon_prob [64], const std::vector <const char *> & stop_codon);
   void  Cumulative_Score
       (const std::string & s, std::vector <double> & score, int frame)  const;
   void  Cumulative_Score_String
       (char * string, int len, int frame, double * score);
   void  Display
       (FILE * fp);
   void  Frame_Score
       (const std::string & s, std::vector <double> & score, int frame)  const;
   void  Full_Window_Distrib
       (char * string, int frame, float * dist);
   double  Full_Window_Prob
       (const char * string, int frame)  const;
   void  Input
       (FILE * fp);
   void  Output
       (FILE * fp, bool binary_form);
   void  Output_Node
       (FILE * fp, ICM_Score_Node_t * node, int id, int frame, bool binary_form);
   double  Partial_Window_Prob
       (int predict_pos, const char * string, int frame)  const;
   void  Read
       (char * path);
   double  Score_String
       (const char * string, int len, int frame)  const;
   void  Set_Label_String
       (char * label, int id, int frame);
   void  Write_Header
       (FILE * fp, bool binary_form);
   void Copy
       (ICM_t & icm);

   // ---- additions ---------------------------------------------------------
   //  Parse an .icm stream without exiting: returns false and sets  err .
   bool  Try_Input
       (FILE * fp, std::string & err);
   //  Flat copies of the tables in the layout gmg_model_upload() takes.
   void  Export_Tables
       (std::vector <short> & mip, std::vector <float> & prob4)  const;
   int  Get_Model_Depth  (void)  const  { return  model_depth; }
   int  Get_Num_Nodes  (void)  const  { return  num_nodes; }
   bool  Is_Empty  (void)  const  { return  empty; }
   //  Device mirror of the tables (uploaded on first use, dropped whenever the
   //  tables change through this class).  Code that edits  score  directly
   //  (a training subclass) must call Invalidate_Device_Mirror() afterwards.
   const gmg_model *  Device_Model  (void)  const;
   void  Invalidate_Device_Mirror  (void)  const;

  private:
   mutable gmg_model  * dev_mirror;
   void  Free_Tables  (void);
   void  Alloc_Tables  (void);
   void  Build_From_Codon_Probs
       (double codon_prob [64], const std::vector <const char *> & stop_codon, const char * who);
  };


//  src/ICM/icm.hh:183-213.  Same constructor and Train_Model as the reference's training class, so that
//  src/ICM/build-icm.cc recompiles against this header unchanged.  The pair counting of every tree level
//  (Count_Char_Pairs, Count_Char_Pairs_Restricted, Get_Training_Node; src/ICM/icm.cc:1190-1256,1841-1870) runs on
//  the device through gmg_trainer_* (include/gmg.h) -- there is no CPU counting path; choosing each node's context
//  position and the chi-squared interpolation (src/ICM/icm.cc:1061-1186,1260-1352) are per-node host arithmetic as
//  in the reference (icm_train.cc).  The reference's per-node count tables (ICM_Training_Node_t, 46 MB for the
//  default shape) are not kept: a level's counts live in HBM while that level is built.
class  ICM_Training_t  :  public ICM_t
  {
  public:
   ICM_Training_t
       (int m = DEFAULT_MODEL_LEN,
        int d = DEFAULT_MODEL_DEPTH,
        int p = DEFAULT_PERIODICITY);
   ~ ICM_Training_t
       ();

   void  Train_Model
       (const std::vector <char *> & data);

   // ---- additions ---------------------------------------------------------
   //  Train_Model without the exit: returns false and sets  err  when the device layer fails.
   bool  Try_Train_Model
       (const char * const * data, int string_ct, std::string & err);
  };


int  Subscript
    (char ch);

#endif
